"""CPU restatement of NeRFace's ray-marching hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *oracle*: a plain torch-on-CPU restatement (fp32 by default, fp64 on
request) of the reference algorithm, used exclusively as the checker by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.  Nothing under
``4d-facial-avatars_amd/`` imports it; the product path is the HIP library and fails loudly
when that library is missing.

Parity pinning: the reference ships no tests/golden vectors for this path, so
the oracle is pinned against the reference *itself*: ``oracle/make_golden.py`` imports the
unmodified reference from /root/reference (inside the build container only), runs it on seeded
synthetic inputs and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this
restatement against those fixtures (bit-exact in fp32 on the same torch build) and, when
/root/reference is present, against the live reference.

File:line citations are relative to /root/reference/nerface_code/nerf-pytorch/ :
  H = nerf/nerf_helpers.py   V = nerf/volume_rendering_utils.py
  T = nerf/train_utils.py    M = nerf/models.py    TN = tiny_nerf.py
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# A1  ray generation                                                         (H:68-123)
# --------------------------------------------------------------------------------------

def ray_bundle(height: int, width: int, intrinsics, c2w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pixel (h, w) -> ray origin/direction, both (H, W, 3).  Follows H:96-123.

    ``intrinsics`` = [fx, fy, cx_rel, cy_rel] (python/numpy doubles); a scalar focal length
    takes the H:109-110 fallback [f, f, .5, .5].  Scalars enter the fp32 tensor arithmetic as
    python numbers, exactly like the reference (double product ``W*cx`` rounded once).
    """
    intr = np.atleast_1d(np.asarray(intrinsics.detach().cpu().numpy() if torch.is_tensor(intrinsics) else intrinsics, dtype=np.float64))
    if intr.shape[0] < 4:
        f = float(intr[0])
        fx, fy, cx, cy = f, f, 0.5, 0.5
    else:
        fx, fy, cx, cy = (float(v) for v in intr[:4])
    dt = c2w.dtype
    col = torch.arange(width, dtype=dt).view(1, width).expand(height, width)   # ii[h, w] = w
    row = torch.arange(height, dtype=dt).view(height, 1).expand(height, width)  # jj[h, w] = h
    d = torch.stack(((col - width * cx) / fx, -(row - height * cy) / fy, -torch.ones_like(col)), dim=-1)
    rot = c2w[:3, :3]
    rd = (d.unsqueeze(-2) * rot).sum(dim=-1)           # rd_i = sum_j d_j R[i, j]     (H:119-121)
    ro = c2w[:3, -1].expand(rd.shape)                   # (H:122)
    return ro, rd


# --------------------------------------------------------------------------------------
# A5  positional encoding                                                    (H:195-239)
# --------------------------------------------------------------------------------------

def posenc(x: torch.Tensor, n_freq: int, include_input: bool = True) -> torch.Tensor:
    """[x?, sin(x f0), cos(x f0), sin(x f1), ...] with f_k = 2**k (log sampling, H:215-221)."""
    parts = [x] if include_input else []
    for k in range(n_freq):
        f = float(2.0 ** k)
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return parts[0] if len(parts) == 1 else torch.cat(parts, dim=-1)


# --------------------------------------------------------------------------------------
# A6  the paper MLP                                                          (M:189-261)
# --------------------------------------------------------------------------------------

PAPER_KEYS = (
    [f"layers_xyz.{i}.{p}" for i in range(6) for p in ("weight", "bias")]
    + [f"{n}.{p}" for n in ("fc_feat", "fc_alpha") for p in ("weight", "bias")]
    + [f"layers_dir.{i}.{p}" for i in range(4) for p in ("weight", "bias")]
    + [f"fc_rgb.{p}" for p in ("weight", "bias")]
)

PAPER_SHAPES = {
    "layers_xyz.0.weight": (256, 171), "layers_xyz.1.weight": (256, 256), "layers_xyz.2.weight": (256, 256),
    "layers_xyz.3.weight": (256, 427), "layers_xyz.4.weight": (256, 256), "layers_xyz.5.weight": (256, 256),
    "fc_feat.weight": (256, 256), "fc_alpha.weight": (1, 256),
    "layers_dir.0.weight": (128, 280), "layers_dir.1.weight": (128, 128), "layers_dir.2.weight": (128, 128),
    "layers_dir.3.weight": (128, 128), "fc_rgb.weight": (3, 128),
}


def init_paper_params(seed: int, dtype=torch.float32, boost=True) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights with nn.Linear-style uniform(-1/sqrt(in), 1/sqrt(in)) init and a density boost.
    boost=True ("hard" family): fc_alpha.weight*1000, fc_alpha.bias=5, fc_rgb.weight*10 -- sharp surfaces, d sigma/dz ~ 1e5,
    a stress test of the resampling sensitivity.  boost="survey" ("soft" family): SURVEY §8(d)'s head, fc_alpha.weight*40,
    fc_alpha.bias=0.5, fc_rgb.weight*10 -- densities of a few units, on which the tight per-stage tolerances of §8(d)
    (rgb 2e-5, weights 1e-5, z_samples 1e-5) are enforced.  (Own generator; not meant to equal torch's default init stream.)"""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for k, shp in PAPER_SHAPES.items():
        bound = 1.0 / math.sqrt(shp[1])
        out[k] = ((torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
        out[k.replace("weight", "bias")] = ((torch.rand(shp[0], generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
    if boost == "survey":
        out["fc_alpha.weight"] = out["fc_alpha.weight"] * 40.0
        out["fc_alpha.bias"] = torch.full_like(out["fc_alpha.bias"], 0.5)
        out["fc_rgb.weight"] = out["fc_rgb.weight"] * 10.0
    elif boost:
        out["fc_alpha.weight"] = out["fc_alpha.weight"] * 1000.0
        out["fc_alpha.bias"] = torch.full_like(out["fc_alpha.bias"], 5.0)
        out["fc_rgb.weight"] = out["fc_rgb.weight"] * 10.0
    return out


def _lin(x, p, name):
    return torch.addmm(p[name + ".bias"], x, p[name + ".weight"].t())


def paper_mlp(p: Dict[str, torch.Tensor], x87: torch.Tensor, expr: torch.Tensor, latent: torch.Tensor,
              masks: Optional[Sequence[torch.Tensor]] = None, acts: Optional[list] = None) -> torch.Tensor:
    """ConditionalBlendshapePaperNeRFModel.forward (M:236-261): (P, 87) -> (P, 4) = [rgb_raw, sigma_raw].

    x0 = [pe_xyz(63) | expr*1/3 (76) | latent (32)]; 3x(Linear+ReLU); skip-concat [x0 | h] at layer 3;
    3x(Linear+ReLU); feat = fc_feat(h) (no activation); sigma = fc_alpha(feat) (Q2: reads feat);
    [feat | pe_dir(24)] -> layers_dir.0..2 (+ReLU) (Q3: layers_dir.3 unused); rgb = fc_rgb.

    Test hooks: `masks` (9 boolean tensors) replaces each ReLU by a multiplication with a given mask, so that a
    gradient comparison is not polluted by units whose pre-activation sits within rounding of zero; `acts`
    (a list) collects the 9 post-ReLU activations and feat.
    """
    n = x87.shape[0]
    xyz, dirs = x87[:, :63], x87[:, 63:]
    e = (expr * 1 / 3).reshape(1, -1).repeat(n, 1)              # true division, M:241
    l = latent.reshape(1, -1).repeat(n, 1)
    x0 = torch.cat((xyz, e, l), dim=1)
    k = [0]

    def act(v):
        out = torch.relu(v) if masks is None else v * masks[k[0]].to(v.dtype)
        k[0] += 1
        if acts is not None:
            acts.append(out)
        return out

    h = x0
    for i in range(6):
        h = act(_lin(torch.cat((x0, h), dim=-1) if i == 3 else h, p, f"layers_xyz.{i}"))
    feat = _lin(h, p, "fc_feat")
    if acts is not None:
        acts.append(feat)
    sigma = _lin(feat, p, "fc_alpha")
    h = act(_lin(torch.cat((feat, dirs), dim=-1), p, "layers_dir.0"))
    h = act(_lin(h, p, "layers_dir.1"))
    h = act(_lin(h, p, "layers_dir.2"))
    rgb = _lin(h, p, "fc_rgb")
    return torch.cat((rgb, sigma), dim=-1)


# --------------------------------------------------------------------------------------
# second model family: ConditionalBlendshapeLearnableCodeNeRFModel                    (M:529-636)
# as every config instantiates it: num_layers=4, hidden_size=256, skip_connect_every left at its default 4
# (the YAML value is never passed, TR:100-109), 10/4 encoding functions, include_input_dir False.
# --------------------------------------------------------------------------------------
LCODE_SHAPES = {
    "layer1.weight": (256, 171), "layers_xyz.0.weight": (256, 256), "layers_xyz.1.weight": (256, 256),
    "layers_xyz.2.weight": (256, 256), "layers_dir.0.weight": (128, 280), "fc_alpha.weight": (1, 256),
    "fc_rgb.weight": (3, 128), "fc_feat.weight": (256, 256),
}
LCODE_KEYS = [k.replace("weight", p) for k in LCODE_SHAPES for p in ("weight", "bias")]


def init_lcode_params(seed: int, dtype=torch.float32, boost=True) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for k, shp in LCODE_SHAPES.items():
        bound = 1.0 / math.sqrt(shp[1])
        out[k] = ((torch.rand(shp, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
        out[k.replace("weight", "bias")] = ((torch.rand(shp[0], generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)
    if boost == "survey":                       # SURVEY 8(d)'s density head, as init_paper_params: the tight per-stage gates apply
        out["fc_alpha.weight"] = out["fc_alpha.weight"] * 40.0
        out["fc_alpha.bias"] = torch.full_like(out["fc_alpha.bias"], 0.5)
        out["fc_rgb.weight"] = out["fc_rgb.weight"] * 10.0
    elif boost:
        out["fc_alpha.weight"] = out["fc_alpha.weight"] * 300.0
        out["fc_alpha.bias"] = torch.full_like(out["fc_alpha.bias"], 5.0)
        out["fc_rgb.weight"] = out["fc_rgb.weight"] * 10.0
    return out


def lcode_mlp(p: Dict[str, torch.Tensor], x87: torch.Tensor, expr: torch.Tensor, latent: torch.Tensor, masks=None,
              acts=None) -> torch.Tensor:
    """M:590-636: x = layer1([xyz | expr*1/3 | latent]) (NO activation); 3 x relu(Linear 256); feat = relu(fc_feat(x));
    alpha = fc_alpha(x) (reads x, not feat); relu(layers_dir.0([feat | dirs])); rgb = fc_rgb.
    Test hooks as in paper_mlp: `masks` (5 boolean tensors: layers_xyz.0..2, fc_feat, layers_dir.0) replaces each ReLU by a
    multiplication with the given mask; `acts` (a list) collects layer1's output and the 5 post-ReLU activations."""
    n = x87.shape[0]
    xyz, dirs = x87[:, :63], x87[:, 63:]
    e = (expr * 1 / 3).reshape(1, -1).repeat(n, 1)
    l = latent.reshape(1, -1).repeat(n, 1)
    k = [0]

    def act(v):
        out = torch.relu(v) if masks is None else v * masks[k[0]].to(v.dtype)
        k[0] += 1
        if acts is not None:
            acts.append(out)
        return out

    x = _lin(torch.cat((xyz, e, l), dim=1), p, "layer1")
    if acts is not None:
        acts.append(x)
    for i in range(3):
        x = act(_lin(x, p, f"layers_xyz.{i}"))
    feat = act(_lin(x, p, "fc_feat"))
    alpha = _lin(x, p, "fc_alpha")
    h = act(_lin(torch.cat((feat, dirs), dim=-1), p, "layers_dir.0"))
    return torch.cat((_lin(h, p, "fc_rgb"), alpha), dim=-1)


def encode_points(ro, rd, z, near: float, far: float, rd_view=None) -> torch.Tensor:
    """run_network's input assembly (T:9-18): pts = ro + rd*z; 'view dirs' = ray_batch[..., -3:] which,
    because the viewdir concat is commented out (T:215-216), is (rd_z, near, far) (Quirk Q1).
    Returns (R*S, 87)."""
    pts = ro[:, None, :] + rd[:, None, :] * z[:, :, None]
    rv = rd if rd_view is None else rd_view            # ablation path (T:81-82, Quirk Q7): encoded direction from other rays
    fake_dirs = torch.stack((rv[:, 2], torch.full_like(rv[:, 2], near), torch.full_like(rv[:, 2], far)), dim=-1)
    dirs = fake_dirs[:, None, :].expand(pts.shape)
    return torch.cat((posenc(pts.reshape(-1, 3), 10, True), posenc(dirs.reshape(-1, 3), 4, False)), dim=-1)


# --------------------------------------------------------------------------------------
# A3  stratified coarse sampler                                              (T:50-78)
# --------------------------------------------------------------------------------------

def coarse_z(n_rays: int, near: float, far: float, n_coarse: int, t_rand: Optional[torch.Tensor], dtype=torch.float32,
             device=None, lindisp: bool = False):
    t = torch.linspace(0.0, 1.0, n_coarse, dtype=dtype, device=device)
    nr = torch.full((n_rays, 1), near, dtype=dtype, device=device)
    fr = torch.full((n_rays, 1), far, dtype=dtype, device=device)
    if not lindisp:
        z = nr * (1.0 - t) + fr * t
    else:                                                       # T:65-66: linear in disparity
        z = 1.0 / (1.0 / nr * (1.0 - t) + 1.0 / fr * t)
    if t_rand is not None:                                      # perturb=True (T:69-76)
        mids = 0.5 * (z[:, 1:] + z[:, :-1])
        upper = torch.cat((mids, z[:, -1:]), dim=-1)
        lower = torch.cat((z[:, :1], mids), dim=-1)
        z = lower + (upper - lower) * t_rand
    return z


# --------------------------------------------------------------------------------------
# A8  volume integrator                                                      (V:7-75, H:44-65)
# --------------------------------------------------------------------------------------

def volume_render(raw: torch.Tensor, z: torch.Tensor, rd: torch.Tensor, noise: Optional[torch.Tensor] = None,
                  has_background: bool = True, white_background: bool = False):
    """raw (R,S,4), z (R,S), rd (R,3) -> rgb_map (R,3), disp (R), acc (R), weights (R,S).

    With ``has_background`` the last sample's colour is used raw (no sigmoid, V:29-31) -- the caller has
    already overwritten raw[:, -1, :3] with the background prior (T:95-96).  sigma = relu(raw_a + noise),
    last sigma += 1e-6 (V:52-53), alpha = 1-exp(-sigma*dist), dist_last = 1e10 (V:19-26), scaled by |rd|.
    """
    big = torch.full_like(z[:, :1], 1e10)
    dists = torch.cat((z[:, 1:] - z[:, :-1], big), dim=-1) * rd[:, None, :].norm(p=2, dim=-1)
    if has_background:
        rgb = torch.cat((torch.sigmoid(raw[:, :-1, :3]), raw[:, -1:, :3]), dim=1)
    else:
        rgb = torch.sigmoid(raw[..., :3])
    a = raw[..., 3] if noise is None else raw[..., 3] + noise
    sigma = torch.relu(a)
    sigma = torch.cat((sigma[:, :-1], sigma[:, -1:] + 1e-6), dim=-1)
    alpha = 1.0 - torch.exp(-sigma * dists)
    trans = torch.cumprod(1.0 - alpha + 1e-10, dim=-1)
    trans = torch.cat((torch.ones_like(trans[:, :1]), trans[:, :-1]), dim=-1)   # exclusive (H:59-63)
    w = alpha * trans
    rgb_map = (w[..., None] * rgb).sum(dim=-2)
    depth = (w * z).sum(dim=-1)
    acc = w.sum(dim=-1)
    disp = 1.0 / torch.max(1e-10 * torch.ones_like(depth), depth / acc)
    if white_background:
        rgb_map = rgb_map + (1.0 - acc[..., None])
    return rgb_map, disp, acc, w


# --------------------------------------------------------------------------------------
# A9  inverse-CDF sampler                                                    (H:344-387)
# --------------------------------------------------------------------------------------

def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, n_samples: int, u: Optional[torch.Tensor] = None,
               table: Optional[dict] = None):
    """bins (R,B), weights (R,B-1) -> (R,n_samples).  ``u=None`` is det mode: linspace(0,1,n) incl. 1.0.
    ``table`` (a dict) collects the CDF and the searchsorted indices (H:353, H:368) for the K6 parity test."""
    w = weights + 1e-5
    pdf = w / w.sum(dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, dim=-1)
    cdf = torch.cat((torch.zeros_like(cdf[:, :1]), cdf), dim=-1)
    if u is None:
        u = torch.linspace(0.0, 1.0, steps=n_samples, dtype=w.dtype, device=w.device).expand(cdf.shape[0], n_samples)
    u = u.contiguous()
    idx = torch.searchsorted(cdf.contiguous(), u, right=True)
    if table is not None:
        table.update(cdf=cdf, inds=idx)
    lo = (idx - 1).clamp(min=0)
    hi = idx.clamp(max=cdf.shape[-1] - 1)
    c_lo, c_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    b_lo, b_hi = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    den = c_hi - c_lo
    den = torch.where(den < 1e-5, torch.ones_like(den), den)
    t = (u - c_lo) / den
    return b_lo + t * (b_hi - b_lo)


# --------------------------------------------------------------------------------------
# A2/A3/A7/A10/A11  predict_and_render_radiance                              (T:36-162)
# --------------------------------------------------------------------------------------

def render_rays(p_coarse, p_fine, ro, rd, expr, latent, bg, near: float, far: float, n_coarse: int, n_fine: int,
                t_rand=None, noise_c=None, u=None, noise_f=None, stages: Optional[dict] = None, rd_view=None, mlp=None,
                point_chunk: int = 65536, lindisp: bool = False, white_background: bool = False):
    """Coarse pass -> hierarchical resample -> fine pass.  Returns the 7-tuple of T:162
    (rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f, weights_f[:, -1]).  Random tensors are injected
    (None = deterministic: perturb off / no noise / det sampling).  ``stages`` collects intermediates.
    white_background: the switch both integrator calls receive (T:104, T:150 -> V:71-72)."""
    R = ro.shape[0]
    st = stages if stages is not None else {}
    mlp_fn = mlp if mlp is not None else globals()["paper_mlp"]             # model family (default: the paper model)

    def paper_mlp(p, x, e, l):
        # run_network feeds the model `chunksize` POINTS at a time (T:20-24, get_minibatches over the embedded points)
        if x.shape[0] <= point_chunk:
            return mlp_fn(p, x, e, l)
        return torch.cat([mlp_fn(p, x[k:k + point_chunk], e, l) for k in range(0, x.shape[0], point_chunk)], dim=0)

    z = coarse_z(R, near, far, n_coarse, t_rand, dtype=ro.dtype, device=ro.device, lindisp=lindisp)
    raw = paper_mlp(p_coarse, encode_points(ro, rd, z, near, far, rd_view), expr, latent).reshape(R, n_coarse, 4).clone()
    st["raw_c_mlp"] = raw.clone()
    if bg is not None:
        raw[:, -1, :3] = bg.to(raw.dtype)
    rgb_c, disp_c, acc_c, w_c = volume_render(raw, z, rd, noise_c, has_background=bg is not None, white_background=white_background)
    st.update(z_c=z, w_c=w_c)
    if n_fine <= 0:
        return rgb_c, disp_c, acc_c, None, None, None, w_c[:, -1]
    z_mid = 0.5 * (z[:, 1:] + z[:, :-1])
    z_s = sample_pdf(z_mid, w_c[:, 1:-1], n_fine, u).detach()           # T:124: no gradient through the resampled depths
    z_f, _ = torch.sort(torch.cat((z, z_s), dim=-1), dim=-1)
    raw_f = paper_mlp(p_fine, encode_points(ro, rd, z_f, near, far, rd_view), expr, latent).reshape(R, n_coarse + n_fine, 4).clone()
    st["raw_f_mlp"] = raw_f.clone()
    if bg is not None:
        raw_f[:, -1, :3] = bg.to(raw_f.dtype)
    rgb_f, disp_f, acc_f, w_f = volume_render(raw_f, z_f, rd, noise_f, has_background=bg is not None, white_background=white_background)
    st.update(z_samples=z_s, z_f=z_f, w_f=w_f)
    return rgb_c, disp_c, acc_c, rgb_f, disp_f, acc_f, w_f[:, -1]


# --------------------------------------------------------------------------------------
# A12  loss of the trainer                                                   (TR:355-387)
# --------------------------------------------------------------------------------------

def train_loss(rgb_c, rgb_f, target, latent):
    mse = torch.nn.functional.mse_loss
    return mse(rgb_c[..., :3], target[..., :3]) + mse(rgb_f[..., :3], target[..., :3]) + 10 * 0.0005 * torch.norm(latent)


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    m = float(torch.mean((a.double() - b.double()) ** 2))
    return -10.0 * math.log10(m if m > 0 else 1e-30)


# --------------------------------------------------------------------------------------
# A13  tiny_nerf path (BASELINE config 1)                                    (TN:12-181)
# --------------------------------------------------------------------------------------

def tiny_init_params(seed: int, filter_size: int = 128, n_freq: int = 10, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    dims = [(filter_size, 3 + 3 * 2 * n_freq), (filter_size, filter_size), (4, filter_size)]
    out = {}
    for i, (o, k) in enumerate(dims, start=1):
        b = 1.0 / math.sqrt(k)
        out[f"layer{i}.weight"] = ((torch.rand((o, k), generator=g, dtype=torch.float64) * 2 - 1) * b).to(dtype)
        out[f"layer{i}.bias"] = ((torch.rand(o, generator=g, dtype=torch.float64) * 2 - 1) * b).to(dtype)
    return out


def flex_init_params(seed: int, num_layers: int = 4, hidden_size: int = 128, n_freq: int = 10, dtype=torch.float32):
    """state_dict of FlexibleNeRFModel(num_layers, hidden_size, num_encoding_fn_xyz=n_freq, include_input_xyz=True,
    use_viewdirs=False) (M:351-394): layer1, layers_xyz.0 .. num_layers-2, fc_out; nn.Linear's default range, seeded."""
    g = torch.Generator().manual_seed(seed)
    dims = [("layer1", hidden_size, 3 + 3 * 2 * n_freq)]
    dims += [(f"layers_xyz.{i}", hidden_size, hidden_size) for i in range(num_layers - 1)]
    dims += [("fc_out", 4, hidden_size)]
    out = {}
    for name, o, k in dims:
        b = 1.0 / math.sqrt(k)
        out[f"{name}.weight"] = ((torch.rand((o, k), generator=g, dtype=torch.float64) * 2 - 1) * b).to(dtype)
        out[f"{name}.bias"] = ((torch.rand(o, generator=g, dtype=torch.float64) * 2 - 1) * b).to(dtype)
    return out


def flex_mlp(p, x):
    """FlexibleNeRFModel.forward with use_viewdirs=False and fewer than 6 layers (M:396-422): layer1 WITHOUT activation (M:402),
    every layers_xyz.i followed by ReLU (M:403-410; the skip of M:404-409 needs i = 4), fc_out (M:422)."""
    n_hidden = sum(1 for k in p if k.startswith("layers_xyz.") and k.endswith(".weight"))
    assert n_hidden <= 4, "the skip connection (M:404-409) is not restated"
    h = _lin(x, p, "layer1")
    for i in range(n_hidden):
        h = torch.relu(_lin(h, p, f"layers_xyz.{i}"))
    return _lin(h, p, "fc_out")


def tiny_render(p, height, width, focal, c2w, near, far, n_samples, n_freq=10, jitter: Optional[torch.Tensor] = None):
    """run_one_iter_of_tinynerf (TN:111-159): whole image, coarse only, no background prior,
    no +1e-6, no |rd| scaling (TN:68-107).  `p`: VeryTinyNerfModel's state_dict (TN:162-181) or, with a "fc_out.weight" entry,
    FlexibleNeRFModel's (flex_mlp)."""
    ro, rd = ray_bundle(height, width, np.asarray([float(focal)]), c2w)
    depth = torch.linspace(near, far, n_samples, dtype=ro.dtype)
    if jitter is not None:                                       # TN:46-57
        depth = depth + jitter * (far - near) / n_samples
    else:
        depth = depth.expand(height, width, n_samples)
    pts = ro[..., None, :] + rd[..., None, :] * depth[..., :, None]
    x = posenc(pts.reshape(-1, 3), n_freq, True)
    if "fc_out.weight" in p:
        raw = flex_mlp(p, x).reshape(height, width, n_samples, 4)
    else:
        h = torch.relu(_lin(x, p, "layer1"))
        h = torch.relu(_lin(h, p, "layer2"))
        raw = _lin(h, p, "layer3").reshape(height, width, n_samples, 4)
    sigma = torch.relu(raw[..., 3])
    rgb = torch.sigmoid(raw[..., :3])
    big = torch.full_like(depth[..., :1], 1e10)
    dists = torch.cat((depth[..., 1:] - depth[..., :-1], big), dim=-1)
    alpha = 1.0 - torch.exp(-sigma * dists)
    trans = torch.cumprod(1.0 - alpha + 1e-10, dim=-1)
    trans = torch.cat((torch.ones_like(trans[..., :1]), trans[..., :-1]), dim=-1)
    w = alpha * trans
    return (w[..., None] * rgb).sum(dim=-2), (w * depth).sum(dim=-1), w.sum(dim=-1)


# --------------------------------------------------------------------------------------
# Synthetic scene shared by tests, bench and the golden generator
# --------------------------------------------------------------------------------------

INTRINSICS = np.array([-1481.96352, 1559.67488, 0.565694, 0.413902], dtype=np.float64)
NEAR, FAR = 0.2, 0.8


def frame_pose(f: int, dtype=torch.float32) -> torch.Tensor:
    """Small yaw/pitch about the head, camera at z~0.5 (4x4 cam2world)."""
    a = 0.3 * math.sin(2 * math.pi * f / 100.0)
    b = 0.3 * math.cos(2 * math.pi * f / 100.0) * 0.5
    ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    m = np.eye(4)
    m[:3, :3] = ry @ rx
    m[:3, 3] = [0.02 * math.sin(2 * math.pi * f / 100.0), 0.02 * math.cos(2 * math.pi * f / 100.0), 0.5]
    return torch.tensor(m, dtype=dtype)


def frame_conditioning(f: int, dtype=torch.float32):
    g = torch.Generator().manual_seed(1000 + f)
    expr = (0.5 * torch.randn(76, generator=g, dtype=torch.float64)).to(dtype)
    latent = (0.1 * torch.randn(32, generator=g, dtype=torch.float64)).to(dtype)
    return expr, latent


def synthetic_image(height: int, width: int, seed: int, dtype=torch.float32) -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    return torch.rand((height, width, 3), generator=g, dtype=torch.float64).to(dtype)


# --------------------------------------------------------------------------------------
# eval post-processing (reference eval_transformed_rays.py:84-119, 184-190).  Pinned: oracle/make_golden.py imports the
# unmodified eval script (torchvision/imageio stubbed, oracle/ref_import.py) and stores its outputs in
# tests/golden/eval_post.npz; this restatement reproduces them bit for bit (tests/test_oracle_golden.py).
# --------------------------------------------------------------------------------------

def cast_to_u8(img: torch.Tensor) -> torch.Tensor:
    """cast_to_image: clamp to [0,1]; torchvision's ToPILImage turns a float tensor into bytes with mul(255).byte()."""
    return img.clamp(0.0, 1.0).mul(255).to(torch.uint8)


def normal_map(depthmap: torch.Tensor, intrinsics, weights: Optional[torch.Tensor] = None) -> torch.Tensor:
    """torch_normal_map(depthmap, focal, weights, clean=True), square images (the script's W/H naming is swapped)."""
    n_rows, n_cols = depthmap.shape
    fx, fy, cx, cy = float(intrinsics[0]), float(intrinsics[1]), float(intrinsics[2]) * n_cols, float(intrinsics[3]) * n_rows
    cc = torch.arange(n_cols, dtype=depthmap.dtype).view(1, -1).expand(n_rows, n_cols)
    rr = torch.arange(n_rows, dtype=depthmap.dtype).view(-1, 1).expand(n_rows, n_cols)
    pts = torch.stack((((cc - cx) * depthmap) / fx, -((rr - cy) * depthmap) / fy, depthmap), dim=-1)
    dx = pts[1:, :, :] - pts[:-1, :, :]
    dy = pts[:, 1:, :] - pts[:, :-1, :]
    nrm = torch.cross(dy[:-1, :, :], dx[:, :-1, :], dim=2)
    nrm = nrm / torch.sqrt(torch.sum(nrm * nrm, 2, keepdim=True))
    nrm = nrm * 0.5 + 0.5
    if weights is not None:
        mask = weights[:-1, :-1].unsqueeze(-1).expand(-1, -1, 3)
        nrm = torch.where(mask > 0.22, torch.ones_like(nrm), nrm)
        nrm = (1 - mask) * nrm + mask * torch.ones_like(nrm)
    return (nrm * 255).to(torch.uint8)
