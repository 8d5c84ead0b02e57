"""Numerical model of the split-fp16 arithmetics (TEST INFRASTRUCTURE / experiment; runs the ORACLE, never the product).

    python -m oracle.split_emulation            (on a GPU box: torch fp64 on cuda; ~3 minutes)

Question (profiles/r05_split_products.md): the single fp16 product per weight ("x1": W_hi x_hi, 1044 MFMAs per 32 points, 15.8 ms per
fine launch = 3.0 M rays/s) misses north_star's 1e-4 dB gate because a rounded WEIGHT is the same error at every point.  Does a
first-order correction rescue it?  Modelled here on whole 512 x 512 frames, in float64 with the operands rounded exactly as the kernels
round them (weights x the layer's power-of-two scale -> fp16 `hi` (+ `lo`); activations x 2^4 -> fp16; exact products, wide accumulation):
    x2   : W_hi + W_lo, x_hi                      (the shipped "f16x2": validates the model against the measured kernel)
    x1   : W_hi, x_hi
    x1c  : x1 + bias correction  b_j += sum_k (W - W_hi)_jk mean(x_k), means from a 2k-point sample of the frame (what the range probe
           of the f16 modes already evaluates per frame and model)
    x1e  : x1 with error-feedback rounding of the weights along K (the rounding errors of a row sum to ~0)
    x1ec : both
Reports |dPSNR| against a random target and self-PSNR against the exact fp64 frame, x1000 and x40 density head."""
import math
import sys

import torch

from . import cases as C
from . import nerface_oracle as O

Q = lambda t: t.to(torch.float16).to(torch.float64)                    # round to nearest even, like v_cvt_f16_f32 / the pack kernel
ACT = 16.0


def q_act(x):
    return Q((x * ACT).clamp(-65504.0, 65504.0)) / ACT


def layer_scale(w):
    amax = float(w.abs().max())
    if not amax > 0:
        return 1.0
    k = math.frexp(amax)[1]
    return 2.0 ** max(-24, min(40, 14 - k))


def quant_rows_feedback(ws):
    """fp16 rounding of the scaled weights with the rounding error of column k carried into column k + 1 of the same row."""
    out = torch.empty_like(ws)
    carry = torch.zeros(ws.shape[0], dtype=ws.dtype, device=ws.device)
    for k in range(ws.shape[1]):
        v = ws[:, k] + carry
        q = Q(v)
        out[:, k] = q
        carry = v - q
    return out


class Emu:
    """One network (parameter dict p, float64 on the device) in one arithmetic."""

    def __init__(self, p, mode, w22=()):
        """mode: x2 | x1 | x1c | x1e | x1ec; w22: layer names that keep 22-bit weights whatever the mode (mixed arithmetics)."""
        self.p, self.mode, self.w22 = p, mode, set(w22)
        self.corr = {}
        two_w = mode == "x2"
        fb = "e" in mode[2:]
        # quantised parts: (name, column slice of the layer input that goes through the MFMAs)
        self.qcols = {"layers_xyz.0": slice(0, 63), "layers_xyz.3": None, "layers_dir.0": None}
        self.wq = {}
        for name in [f"layers_xyz.{i}" for i in range(6)] + ["fc_feat", "layers_dir.0", "layers_dir.1", "layers_dir.2", "fc_rgb", "fc_alpha"]:
            w = p[name + ".weight"]
            stream = self.stream_mask(name, w.shape[1])
            ws_all = w[:, stream]
            if name in ("layers_dir.0", "fc_alpha"):                # fc_alpha rides as a tile of layers_dir.0: one scale for both
                s = layer_scale(torch.cat((p["layers_dir.0.weight"][:, self.stream_mask("layers_dir.0", 280)].reshape(-1), p["fc_alpha.weight"].reshape(-1))))
            else:
                s = layer_scale(ws_all)
            hi = (quant_rows_feedback(ws_all * s) if fb else Q(ws_all * s))
            if name in self.w22:
                hi = Q(ws_all * s)
            wq = hi + (Q(ws_all * s - hi) if (two_w or name in self.w22) else 0.0)
            full = w.clone()
            full[:, stream] = wq / s
            self.wq[name] = full

    @staticmethod
    def stream_mask(name, n_in):
        m = torch.ones(n_in, dtype=torch.bool)
        if name == "layers_xyz.0":
            m[63:] = False                                           # expression / latent columns are folded into the bias in f32
        elif name == "layers_xyz.3":
            m[63:171] = False
        elif name == "layers_dir.0":
            for f in range(4):
                for sc in range(2):
                    m[256 + 6 * f + 3 * sc + 1] = False              # PE(near), PE(far): folded
                    m[256 + 6 * f + 3 * sc + 2] = False
        return m

    def lin(self, name, x, xq):
        """exact columns take x, streamed columns take the rounded operand xq"""
        w, wq = self.p[name + ".weight"], self.wq[name]
        m = self.stream_mask(name, w.shape[1]).to(x.device)
        xin = torch.where(m, xq, x)
        y = torch.addmm(self.p[name + ".bias"], xin, wq.t())
        if name in self.corr:
            y = y + self.corr[name]
        return y

    def forward(self, x87, expr, latent, means=None):
        n = x87.shape[0]
        e = (expr * 1 / 3).reshape(1, -1).repeat(n, 1)
        l = latent.reshape(1, -1).repeat(n, 1)
        x0 = torch.cat((x87[:, :63], e, l), dim=1)
        rec = lambda name, t: means.__setitem__(name, t.mean(0)) if means is not None else None
        h = x0
        for i in range(6):
            xin = torch.cat((x0, h), dim=-1) if i == 3 else h
            rec(f"layers_xyz.{i}", xin)
            h = torch.relu(self.lin(f"layers_xyz.{i}", xin, q_act(xin)))
        rec("fc_feat", h)
        feat = self.lin("fc_feat", h, q_act(h))
        rec("fc_alpha", feat)
        sigma = self.lin("fc_alpha", feat, q_act(feat))
        xin = torch.cat((feat, x87[:, 63:]), dim=-1)
        rec("layers_dir.0", xin)
        h = torch.relu(self.lin("layers_dir.0", xin, q_act(xin)))
        for i in (1, 2):
            rec(f"layers_dir.{i}", h)
            h = torch.relu(self.lin(f"layers_dir.{i}", h, q_act(h)))
        rec("fc_rgb", h)
        rgb = self.lin("fc_rgb", h, q_act(h))
        return torch.cat((rgb, sigma), dim=-1)

    def calibrate(self, means):
        """first-order bias correction: (W - W_q) mean(x) per layer, from the sampled means of the EXACT activations"""
        for name, mu in means.items():
            self.corr[name] = ((self.p[name + ".weight"] - self.wq[name]) @ mu).reshape(1, -1)


def render(c, dev, emus=None, rays=None, chunk=4096, capture=None):
    pc = {k: v.to(device=dev, dtype=torch.float64) for k, v in c["p_coarse"].items()}
    pf = {k: v.to(device=dev, dtype=torch.float64) for k, v in c["p_fine"].items()}
    expr, lat = c["expr"].to(device=dev, dtype=torch.float64), c["latent"].to(device=dev, dtype=torch.float64)
    ro, rd, bg = rays

    def mlp(p, x, e, l):
        which = "c" if p is pc else "f"
        if capture is not None:
            capture.setdefault(which, []).append(x)
        if emus is None:
            return O.paper_mlp(p, x, e, l)
        return emus[which].forward(x, e, l)
    parts = []
    with torch.no_grad():
        for k in range(0, ro.shape[0], chunk):
            f = lambda t: t[k:k + chunk].to(device=dev, dtype=torch.float64)
            parts.append(O.render_rays(pc, pf, f(ro), f(rd), expr, lat, f(bg), O.NEAR, O.FAR, 64, 128, mlp=mlp)[3])
    return torch.cat(parts, dim=0), pc, pf, expr, lat


MIXES = {  # name: (layers that keep W_lo, MFMAs per 32 points of the kernel that would implement it)
    "feat+alpha": (("fc_feat", "fc_alpha"), 1152),
    "h5+feat+alpha": (("layers_xyz.5", "fc_feat", "fc_alpha"), 1280),
    "h3..5+feat+alpha": (("layers_xyz.3", "layers_xyz.4", "layers_xyz.5", "fc_feat", "fc_alpha"), 1568),
    "trunk+feat+alpha (dir layers x1)": (tuple(f"layers_xyz.{i}" for i in range(6)) + ("fc_feat", "fc_alpha"), 1856),
}


def main_mixes():
    """Mixed arithmetics: 11-bit activations everywhere, 22-bit weights only on the named layers, 11-bit (+ bias correction) on the rest."""
    dev = torch.device("cuda:0")
    H = W = 512
    psnr = lambda a, b: -10.0 * float(torch.log10(torch.mean((a - b) ** 2)))
    for family, case in (("x1000 head", "eval_det_64_128"), ("x40 head", "soft_eval_det_64_128")):
        c = C.build_case(case)
        ro, rd = O.ray_bundle(H, W, O.INTRINSICS, O.frame_pose(c["frame"]))
        bg, tgt = O.synthetic_image(H, W, 7).reshape(-1, 3), O.synthetic_image(H, W, 11).reshape(-1, 3).to(dev).double()
        rays = (ro.reshape(-1, 3), rd.reshape(-1, 3), bg)
        ref, pc, pf, expr, lat = render(c, dev, None, rays)
        cap = {}
        render(c, dev, None, tuple(t[::1024] for t in rays), capture=cap)
        for name, (w22, n_mfma) in MIXES.items():
            for mode in ("x1", "x1c"):
                emus = {"c": Emu(pc, mode, w22), "f": Emu(pf, mode, w22)}
                if mode == "x1c":
                    for which, p in (("c", pc), ("f", pf)):
                        means = {}
                        with torch.no_grad():
                            Emu(p, "x2").forward(torch.cat(cap[which], 0)[::8], expr, lat, means=means)
                        emus[which].calibrate({k: v for k, v in means.items() if k not in w22})
                img = render(c, dev, emus, rays)[0]
                print(f"[{family}] W22 on {name} ({n_mfma} MFMAs), rest {mode}: |dPSNR| = {abs(psnr(img, tgt) - psnr(ref, tgt)):.2e} dB, "
                      f"self-PSNR {psnr(img, ref):.1f} dB", flush=True)


def main():
    dev = torch.device("cuda:0")
    H = W = 512
    psnr = lambda a, b: -10.0 * float(torch.log10(torch.mean((a - b) ** 2)))
    for family, case in (("x1000 head", "eval_det_64_128"), ("x40 head", "soft_eval_det_64_128")):
        c = C.build_case(case)
        ro, rd = O.ray_bundle(H, W, O.INTRINSICS, O.frame_pose(c["frame"]))
        bg, tgt = O.synthetic_image(H, W, 7).reshape(-1, 3), O.synthetic_image(H, W, 11).reshape(-1, 3).to(dev).double()
        rays = (ro.reshape(-1, 3), rd.reshape(-1, 3), bg)
        ref, pc, pf, expr, lat = render(c, dev, None, rays)
        # calibration sample: every 1024th ray (256 rays), the networks' own inputs on those rays, every 8th point (the range probe's sample)
        cap = {}
        sub = tuple(t[::1024] for t in rays)
        render(c, dev, None, sub, capture=cap)
        for mode in ("x2", "x1", "x1c", "x1e", "x1ec"):
            emus = {"c": Emu(pc, mode), "f": Emu(pf, mode)}
            if mode.endswith("c"):
                for which, p in (("c", pc), ("f", pf)):
                    means = {}
                    with torch.no_grad():
                        Emu(p, "x2").forward(torch.cat(cap[which], 0)[::8], expr, lat, means=means)      # means of (near-)exact activations
                    emus[which].calibrate(means)
            img = render(c, dev, emus, rays)[0]
            print(f"[{family}] {mode:5s}: |dPSNR| = {abs(psnr(img, tgt) - psnr(ref, tgt)):.2e} dB, self-PSNR {psnr(img, ref):.1f} dB, "
                  f"mean d rgb {float((img - ref).mean()):+.2e}, max|d rgb| {float((img - ref).abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    main_mixes() if sys.argv[1:] == ["mixes"] else main()
