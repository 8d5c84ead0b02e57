"""A/B of the exact-f32 training kernels of the paper model: round-3 (line-wide deferred saves + ReLU bit masks in the forward
and the dX chain, shared-panel weight-gradient kernel) against round-2 (nf_debug_legacy_train(1)), same inputs, same process.

Checks that raw, every saved section and every dZ section are BIT-IDENTICAL between the two and that the 26 gradients + d latent
agree to rounding (the weight-gradient kernel slices the points differently, so the slab sums associate differently), and times the forward-with-saves call and the backward call (chain + dW + reduce + unpack) with HIP events on torch's stream.

    python tools/ab_train_f32.py [--rays 2048] [--samples 64 128] [--iters 20] [--json out.json]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))

from nerf import _hip as H  # noqa: E402
from nerf import models, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=2048)
    ap.add_argument("--samples", type=int, nargs="+", default=[64, 128])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--json", default=None)
    ap.add_argument("--odd", action="store_true", help="also run a ragged size (2047 rays x 127 samples)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = H.lib()
    dbg = lib.nf_debug_legacy_train
    dbg.restype, dbg.argtypes = None, [C.c_int]
    torch.manual_seed(0)
    model = models.ConditionalBlendshapePaperNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True,
                                                       include_input_dir=False, use_viewdirs=True, include_expression=True,
                                                       latent_code_dim=32).to(dev)
    hw = model.hip_weights()
    packed, packed_t = hw.get(), hw.get_t()
    expr = torch.randn(76, device=dev) * 0.3
    latent = torch.randn(32, device=dev) * 0.1
    cond = ops.paper_condition(packed, expr, latent, 0.2, 0.8)
    out = {}
    cases = [(args.rays, s) for s in args.samples] + ([(2047, 127)] if args.odd else [])
    for n_rays, n_s in cases:
        n = n_rays * n_s
        ro = torch.randn(n_rays, 3, device=dev) * 0.1
        rd = torch.nn.functional.normalize(torch.randn(n_rays, 3, device=dev), dim=-1)
        z = torch.sort(torch.rand(n_rays, n_s, device=dev) * 0.6 + 0.2, dim=-1)[0]
        d_raw = torch.randn(n_rays, n_s, 4, device=dev) * torch.rand(n_rays, n_s, 1, device=dev) ** 4
        res = {}
        for legacy in (1, 0):
            dbg(legacy)
            raw = torch.empty(n_rays, n_s, 4, device=dev)
            saved = torch.zeros(lib.nf_paper_saved_floats(n), device=dev)
            ws_floats = lib.nf_paper_bwd_workspace_floats(n)
            ws = torch.zeros(ws_floats, device=dev)
            flat = torch.empty(lib.nf_paper_grad_floats(), device=dev)

            def fwd():
                H.check(lib.nf_paper_mlp_fwd_train(H.ptr(packed), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd), H.ptr(z), n_rays, n_s,
                                                   H.ptr(raw), H.ptr(saved), H.stream_ptr(dev)), "fwd_train")

            def bwd():
                H.check(lib.nf_paper_mlp_bwd(H.ptr(packed), H.ptr(packed_t), H.ptr(cond), H.ptr(saved), H.ptr(d_raw), n_rays, n_s,
                                             H.ptr(ws), ws_floats, H.ptr(flat), H.stream_ptr(dev)), "bwd")

            def timed(fn):
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for _ in range(args.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / args.iters

            t_f = timed(fwd)
            t_b = timed(bwd)
            res[legacy] = dict(raw=raw.clone(), saved=saved[:2256 * n].clone(), dz=ws[:2176 * n].clone(), flat=flat.clone(), t_f=t_f, t_b=t_b)
        dbg(0)
        same = {k: bool(torch.equal(res[0][k], res[1][k])) for k in ("raw", "saved", "dz")}
        worst = {k: float((res[0][k] - res[1][k]).abs().max()) for k in ("raw", "saved", "dz", "flat")}
        grad_rel = float((res[0]["flat"].double() - res[1]["flat"].double()).norm() / res[1]["flat"].double().norm())
        same["grads_to_rounding"] = grad_rel < 2e-6
        key = f"{n_rays}x{n_s}"
        out[key] = dict(points=n, bit_identical=same, max_abs_diff=worst, grads_rel_l2=grad_rel,
                        fwd_save_ms=dict(r02=res[1]["t_f"], r03=res[0]["t_f"]), bwd_ms=dict(r02=res[1]["t_b"], r03=res[0]["t_b"]))
        print(key, json.dumps(out[key]))
        del res
        torch.cuda.empty_cache()
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)
    ok = all(all(v["bit_identical"].values()) for v in out.values())
    print("A/B", "OK: forward / dZ bit-identical, gradients equal to rounding" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
