"""Tensor-level wrappers over the C ABI (one function per entry point of include/nerface_hip.h).

These allocate outputs with torch, pass raw device pointers + the current HIP stream, and translate
non-zero return codes into RuntimeError.  No numerical work happens in Python.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional, Sequence

import numpy as np
import torch

from . import _hip as H

PAPER_KEYS = (
    [f"layers_xyz.{i}.{p}" for i in range(6) for p in ("weight", "bias")]
    + [f"{n}.{p}" for n in ("fc_feat", "fc_alpha") for p in ("weight", "bias")]
    + [f"layers_dir.{i}.{p}" for i in range(4) for p in ("weight", "bias")]
    + [f"fc_rgb.{p}" for p in ("weight", "bias")]
)


# Arithmetic of the MLP GEMMs: "f32" = exact-f32 MFMA (the library default and the arithmetic of the reference);
# "bf16x3" = split-bf16, three bf16 MFMAs per product with f32 accumulation (~2^-16 relative per product, 3x faster).
# The switch applies to inference AND to a training step: under "bf16x3" the training forward, the dX chain and the
# weight-gradient GEMMs all run on the split-bf16 kernels (paper_mlp_bwd(..., exact_dw=True) keeps the dW GEMMs exact).
# "f16x3" = split-fp16: the same three-MFMA scheme on fp16 pairs (22 operand bits, per-layer power-of-two weight scales,
# per-point block-floating-point gradient scales): fp32-class accuracy at the bf16x3 speed, for inference and for all three training GEMM kernels
# of both model families.
# "f16x2" (round 5): INFERENCE only -- the split-fp16 kernel with two products per weight (W_hi x_hi + W_lo x_hi: activations enter with
# fp16's 11 bits, weights keep 22), a third fewer MFMAs than "f16x3"; whole frames stay within 1e-4 dB PSNR of the reference (measured
# 2e-6 .. 5.6e-5 dB, median 9e-6, over 21 frames), per-point outputs carry 2^-12 relative rounding.  A training step under "f16x2" raises (train with "f16x3").
_VALID_PRECISIONS = ("f32", "bf16x3", "f16x3", "f16x2")
F16_MODES = ("f16x3", "f16x2")           # arithmetics that run on the scaled fp16 weight stream (range probe + range flag apply)
INFERENCE_ONLY_PRECISIONS = ("f16x2",)
_mlp_precision = os.environ.get("NERFACE_MLP_PRECISION", "f32")


def set_mlp_precision(mode: str) -> None:
    global _mlp_precision
    if mode not in _VALID_PRECISIONS:
        raise ValueError(f"precision must be one of {_VALID_PRECISIONS}")
    _mlp_precision = mode


def get_mlp_precision() -> str:
    return _mlp_precision


def _c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------- tables
_LINSPACE_CACHE = {}


def linspace01(n: int, device) -> torch.Tensor:
    """torch.linspace(0, 1, n) as the reference calls it (T:50-55, H:357-360).  The table is produced by
    torch's own CPU kernel once per (n, device) and cached on the device, so that the deterministic
    depths / CDF abscissae carry exactly torch's rounding."""
    key = (int(n), str(device))
    t = _LINSPACE_CACHE.get(key)
    if t is None:
        t = torch.linspace(0.0, 1.0, int(n), dtype=torch.float32).to(device)
        _LINSPACE_CACHE[key] = t
    return t


# ---------------------------------------------------------------------------------------- K1
def ray_bundle(height: int, width: int, fx: float, fy: float, cx: float, cy: float, c2w: torch.Tensor):
    c2w = c2w if (c2w.dtype == torch.float32 and c2w.stride(-1) == 1) else c2w.to(torch.float32).contiguous()
    if not c2w.is_cuda:
        raise RuntimeError("get_ray_bundle (MI355X build): tform_cam2world must be on a ROCm device")
    if c2w.dim() != 2 or c2w.shape[0] < 3 or c2w.shape[1] < 4:
        raise ValueError("tform_cam2world must be (3|4, 4)")
    dev = c2w.device
    ro = torch.empty((height, width, 3), dtype=torch.float32, device=dev)
    rd = torch.empty((height, width, 3), dtype=torch.float32, device=dev)
    # W*cx and H*cy are formed in double and rounded once, as python-scalar operands are in the reference
    cx_w = float(np.float32(np.float64(width) * np.float64(cx)))
    cy_h = float(np.float32(np.float64(height) * np.float64(cy)))
    with torch.cuda.device(dev):
        H.check(H.lib().nf_ray_bundle(height, width, float(np.float32(fx)), float(np.float32(fy)), cx_w, cy_h,
                                      H.ptr(c2w), int(c2w.stride(0)), H.ptr(ro), H.ptr(rd), H.stream_ptr(dev)),
                "nf_ray_bundle")
    return ro, rd


# ---------------------------------------------------------------------------------------- K0
def weighted_choice(weights: torch.Tensor, n: int, u: Optional[torch.Tensor] = None, check: bool = False) -> torch.Tensor:
    """np.random.choice(len(weights), size=n, replace=False, p=weights / weights.sum()) on the device (TR:320-322): n distinct int64
    indices in ascending order (a seed reproduces the batch element for element).  u: the uniform numbers to use (len(weights), in [0, 1)); default torch.rand on the device, i.e.
    torch's generator and seeds.  check=True reads the "fewer than n positive weights" flag back (host sync) and raises ValueError
    like numpy does."""
    w = _c(weights.reshape(-1))
    dev = H.require_device(w)
    n_items = int(w.numel())
    if not 0 <= n <= n_items:
        raise ValueError("Cannot take a larger sample than population when replace is False")
    if u is None:
        u = torch.rand(n_items, dtype=torch.float32, device=dev)
    u = _c(u.reshape(-1))
    H.require_device(u)
    if u.numel() != n_items:
        raise ValueError("u must hold one uniform number per weight")
    lib = H.lib()
    ws_bytes = int(lib.nf_weighted_choice_workspace_bytes())
    ws = torch.empty(ws_bytes // 4, dtype=torch.int32, device=dev)
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        H.check(lib.nf_weighted_choice(H.ptr(w), H.ptr(u), n_items, int(n), H.ptr(idx), H.ptr(ws), ws_bytes, H.stream_ptr(dev)),
                "nf_weighted_choice")
    if check and n > 0 and int(ws[5].item()):
        raise ValueError("Fewer non-zero entries in p than size")
    return idx


def ray_batch(height: int, width: int, fx: float, fy: float, cx: float, cy: float, c2w: torch.Tensor, sel: torch.Tensor,
              image: Optional[torch.Tensor] = None, background: Optional[torch.Tensor] = None, check: bool = False):
    """Rays of the selected pixels only (sel (n, 2) int64 {row, col}, or (n,) flat pixel indices row * W + col) -- bit-identical to
    get_ray_bundle(...)[sel[:, 0], sel[:, 1]]
    -- plus the target pixels of `image` (H, W, C) and the background prior (H, W, 3) at the same pixels, in one launch
    (TR:302, 325-330).  Returns (ro, rd, target | None, bg | None).  check=True reads the out-of-range flag back (host sync)."""
    c2w = c2w if (c2w.dtype == torch.float32 and c2w.stride(-1) == 1) else c2w.to(torch.float32).contiguous()
    if sel.dtype != torch.int64 or not (sel.dim() == 1 or (sel.dim() == 2 and sel.shape[1] == 2)):
        raise ValueError("sel must be an (n, 2) int64 tensor of {row, col} or an (n,) int64 tensor of flat pixel indices")
    sel, image, background = sel.contiguous(), _c(image), _c(background)
    dev = H.require_device(c2w, image, background)
    if sel.device != dev:
        raise RuntimeError("ray_batch: select_inds must live on the device of the pose")
    n = int(sel.shape[0])
    if image is not None and (image.dim() != 3 or tuple(image.shape[:2]) != (height, width) or image.dtype != torch.float32):
        raise ValueError("image must be a float32 (H, W, C) tensor")
    if background is not None and (tuple(background.shape) != (height, width, 3) or background.dtype != torch.float32):
        raise ValueError("background must be a float32 (H, W, 3) tensor")
    ch = int(image.shape[2]) if image is not None else 0
    ro = torch.empty((n, 3), dtype=torch.float32, device=dev)
    rd = torch.empty((n, 3), dtype=torch.float32, device=dev)
    target = torch.empty((n, ch), dtype=torch.float32, device=dev) if image is not None else None
    bg = torch.empty((n, 3), dtype=torch.float32, device=dev) if background is not None else None
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    cx_w = float(np.float32(np.float64(width) * np.float64(cx)))
    cy_h = float(np.float32(np.float64(height) * np.float64(cy)))
    with torch.cuda.device(dev):
        H.check(H.lib().nf_ray_batch(height, width, float(np.float32(fx)), float(np.float32(fy)), cx_w, cy_h, H.ptr(c2w),
                                     int(c2w.stride(0)), H.ptr(sel), 1 if sel.dim() == 1 else 0, n, H.ptr(image), ch, H.ptr(background),
                                     H.ptr(ro), H.ptr(rd),
                                     H.ptr(target), H.ptr(bg), H.ptr(flag), H.stream_ptr(dev)), "nf_ray_batch")
    if check and int(flag.item()):
        raise IndexError("ray_batch: a selected pixel lies outside the image")
    return ro, rd, target, bg


# ---------------------------------------------------------------------------------------- K2
def sample_coarse(n_rays: int, n_coarse: int, near: float, far: float, device, t_rand: Optional[torch.Tensor] = None,
                  lindisp: bool = False):
    """T:56-76: stratified depths (n_rays, n_coarse); lindisp = the reference's linear-in-disparity spacing (T:65-66)."""
    t_rand = _c(t_rand)
    tv = linspace01(n_coarse, device)
    z = torch.empty((n_rays, n_coarse), dtype=torch.float32, device=device)
    if t_rand is not None:
        H.require_device(t_rand)
        assert tuple(t_rand.shape) == (n_rays, n_coarse)
    with torch.cuda.device(device):
        H.check(H.lib().nf_sample_coarse_ex(n_rays, n_coarse, float(np.float32(near)), float(np.float32(far)), H.ptr(tv),
                                            H.ptr(t_rand), 1 if lindisp else 0, H.ptr(z), H.stream_ptr(device)), "nf_sample_coarse_ex")
    return z


# ---------------------------------------------------------------------------------------- K3
def posenc(x: torch.Tensor, n_freq: int, include_input: bool) -> torch.Tensor:
    x = _c(x)
    dev = H.require_device(x)
    dim = x.shape[-1]
    rows = x.numel() // dim if dim else 0
    width = dim * ((1 if include_input else 0) + 2 * n_freq)
    out = torch.empty(tuple(x.shape[:-1]) + (width,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        H.check(H.lib().nf_posenc(H.ptr(x), rows, dim, n_freq, 1 if include_input else 0, H.ptr(out), H.stream_ptr(dev)),
                "nf_posenc")
    return out


# ---------------------------------------------------------------------------------------- K4
# Packed weight images are cached per (parameter storage, version counter, PACK EPOCH).  The version counter follows ordinary
# in-place updates (optimizer.step() of the default / foreach optimizers, load_state_dict, copy_ under no_grad) -- but NOT every
# writer bumps it: torch's FUSED optimizers (Adam(fused=True)), writes through `p.data` and c10d collectives leave it untouched.
# So every run_one_iter_of_nerf / run_one_iter_of_tinynerf call advances the pack epoch: the images are rebuilt once per
# rendered frame or training step (a few 10-microsecond kernels), and a stale image can never outlive one call.
_PACK_EPOCH = [0]


def bump_pack_epoch() -> None:
    _PACK_EPOCH[0] += 1


def pack_epoch() -> int:
    return _PACK_EPOCH[0]


class PaperWeights:
    """Kernel-ready images of one ConditionalBlendshapePaperNeRFModel on one device, one per kind, each re-packed whenever a
    parameter's version counter or the pack epoch moves (i.e. after optimizer.step() / load_state_dict() / a new frame or step):
      f32 / f32_t      fragment-ordered f32 image for the forward / transposed image for the backward chain
      bf16 / bf16_t    (hi, lo) bf16 streams of the split-bf16 forward / chain
      f16 / f16_t      (hi, lo) fp16 streams + per-layer scales of the split-fp16 forward / chain."""

    # kind -> (size function, pack function, element dtype)
    _KINDS = {
        "f32": ("nf_paper_packed_floats", "nf_paper_pack", torch.float32),
        "f32_t": ("nf_paper_packed_bwd_floats", "nf_paper_pack_bwd", torch.float32),
        "bf16": ("nf_paper_packed_bf16_bytes", "nf_paper_pack_bf16", torch.uint8),
        "bf16_t": ("nf_paper_packed_bwd_bf16_bytes", "nf_paper_pack_bwd_bf16", torch.uint8),
        "f16": ("nf_paper_packed_f16_bytes", "nf_paper_pack_f16", torch.uint8),
        "f16_t": ("nf_paper_packed_bwd_f16_bytes", "nf_paper_pack_bwd_f16", torch.uint8),
    }

    def __init__(self, params: Sequence[torch.Tensor]):
        assert len(params) == H.NF_PAPER_NUM_PARAMS
        self._params = list(params)
        self._cache = {}                # kind -> (signature, buffer)
        self._f16_sticky = None         # range-guard flag of the split-fp16 forward carried across re-packs (0-d int32 on the device)

    def invalidate(self) -> None:
        """Drop every cached image.  The caches follow in-place updates through the parameters' version counters
        (optimizer.step(), load_state_dict(), copy_ under no_grad); writes that bypass the counter -- through `p.data`, or by
        a collective -- are NOT seen: call this (or model.hip_weights().invalidate()) after such a write."""
        for kind, (_, buf) in list(self._cache.items()):
            self._cache[kind] = (None, buf)

    def _signature(self):
        return (_PACK_EPOCH[0],) + tuple((int(p.data_ptr()), int(p._version)) for p in self._params)

    def _get(self, kind: str) -> torch.Tensor:
        sig = self._signature()
        hit = self._cache.get(kind)
        if hit is None or hit[0] != sig:
            size_fn, pack_fn, dtype = self._KINDS[kind]
            dev = H.require_device(*[p.detach() for p in self._params])
            lib = H.lib()
            buf = hit[1] if hit is not None and hit[1].device == dev else torch.empty(getattr(lib, size_fn)(), dtype=dtype, device=dev)
            if kind == "f16" and hit is not None and hit[1] is buf:
                # packing clears the stream's range-guard flag: carry it over first (one tiny device op, no host sync), so that a
                # training run polled every print_every iterations still sees an overflow of any iteration in between
                if self._f16_sticky is None or self._f16_sticky.device != dev:
                    self._f16_sticky = torch.zeros((), dtype=torch.int32, device=dev)
                off = lib.nf_paper_f16_flag_offset()
                self._f16_sticky.bitwise_or_(buf[off:off + 4].view(torch.int32)[0])
            arr = (C.c_void_p * H.NF_PAPER_NUM_PARAMS)(*[int(p.data_ptr()) for p in self._params])
            with torch.cuda.device(dev):
                H.check(getattr(lib, pack_fn)(arr, H.ptr(buf), H.stream_ptr(dev)), pack_fn)
            self._cache[kind] = (sig, buf)
        return self._cache[kind][1]

    def get(self) -> torch.Tensor:
        return self._get("f32")

    def get_t(self) -> torch.Tensor:
        return self._get("f32_t")

    def get_bf16(self) -> torch.Tensor:
        return self._get("bf16")

    def get_bf16_t(self) -> torch.Tensor:
        return self._get("bf16_t")

    def get_f16(self) -> torch.Tensor:
        return self._get("f16")

    def get_f16_t(self) -> torch.Tensor:
        return self._get("f16_t")

    def f16_range_flag(self) -> Optional[torch.Tensor]:
        """0-d int32 device tensor: non-zero once a split-fp16 forward of this model produced a non-finite output (an activation
        left fp16's range) -- with the current stream or, carried over the re-packs of a training run, with any earlier one.
        None if the split-fp16 stream was never built."""
        hit = self._cache.get("f16")
        if hit is None:
            return None
        off = H.lib().nf_paper_f16_flag_offset()
        flag = hit[1][off:off + 4].view(torch.int32)[0]
        return flag if self._f16_sticky is None else torch.bitwise_or(flag, self._f16_sticky)


def paper_condition(packed: torch.Tensor, expr: torch.Tensor, latent: torch.Tensor, near: float, far: float) -> torch.Tensor:
    expr, latent = _c(expr.detach()), _c(latent.detach())
    dev = H.require_device(packed, expr, latent)
    if expr.numel() != 76 or latent.numel() != 32:
        raise ValueError("expected a 76-d expression and a 32-d latent code")
    lib = H.lib()
    cond = torch.empty(lib.nf_paper_cond_floats(), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        H.check(lib.nf_paper_condition(H.ptr(packed), H.ptr(expr), H.ptr(latent), float(np.float32(near)),
                                       float(np.float32(far)), H.ptr(cond), H.stream_ptr(dev)), "nf_paper_condition")
    return cond


def paper_mlp_fwd(packed, cond, ro, rd, z, rd_view=None) -> torch.Tensor:
    dev = H.require_device(packed, cond, ro, rd, z, rd_view)
    n_rays, n_samples = z.shape
    raw = torch.empty((n_rays, n_samples, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        H.check(H.lib().nf_paper_mlp_fwd(H.ptr(packed), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view), H.ptr(z),
                                         n_rays, n_samples, H.ptr(raw), H.stream_ptr(dev)), "nf_paper_mlp_fwd")
    return raw


def paper_mlp_fwd_bf16(packed_b, cond, ro, rd, z, rd_view=None) -> torch.Tensor:
    """Split-bf16 (3 x bf16 MFMA, f32 accumulate) forward; same outputs as paper_mlp_fwd to ~1e-5 relative."""
    dev = H.require_device(cond, ro, rd, z, rd_view)
    n_rays, n_samples = z.shape
    raw = torch.empty((n_rays, n_samples, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        H.check(H.lib().nf_paper_mlp_fwd_bf16(H.ptr(packed_b), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view), H.ptr(z),
                                              n_rays, n_samples, H.ptr(raw), H.stream_ptr(dev)), "nf_paper_mlp_fwd_bf16")
    return raw


def paper_mlp_fwd_f16(packed_h, cond, ro, rd, z, rd_view=None) -> torch.Tensor:
    """Split-fp16 (3 x fp16 MFMA on scaled weights, f32 accumulate) forward; fp32-class accuracy (see csrc/nf_mlp_f16.hip)."""
    dev = H.require_device(cond, ro, rd, z, rd_view)
    n_rays, n_samples = z.shape
    raw = torch.empty((n_rays, n_samples, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        H.check(H.lib().nf_paper_mlp_fwd_f16(H.ptr(packed_h), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view), H.ptr(z),
                                             n_rays, n_samples, H.ptr(raw), H.stream_ptr(dev)), "nf_paper_mlp_fwd_f16")
    return raw


def paper_mlp_fwd_f16x2(packed_h, cond, ro, rd, z, rd_view=None) -> torch.Tensor:
    """"f16x2" forward: the split-fp16 kernel with two products per weight (csrc/nf_mlp_f16x2.hip) on the SAME packed image as f16x3."""
    dev = H.require_device(cond, ro, rd, z, rd_view)
    n_rays, n_samples = z.shape
    raw = torch.empty((n_rays, n_samples, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        H.check(H.lib().nf_paper_mlp_fwd_f16x2(H.ptr(packed_h), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view), H.ptr(z),
                                               n_rays, n_samples, H.ptr(raw), H.stream_ptr(dev)), "nf_paper_mlp_fwd_f16x2")
    return raw


def require_trainable_precision() -> None:
    """A training step (need_grad) under an inference-only arithmetic is refused, not silently run on another one."""
    if _mlp_precision in INFERENCE_ONLY_PRECISIONS:
        raise RuntimeError(f'nerf.set_mlp_precision("{_mlp_precision}") is an inference arithmetic: train with "f32", "f16x3" or "bf16x3"')


_f16_train_probe_every = [128]           # training: the f32 range probe runs on a model's first forward and every n-th after it


def set_f16_train_probe_every(n: int) -> None:
    """Cadence of the split-fp16 range probe in training (calls per model; the launcher sets it to its print_every)."""
    _f16_train_probe_every[0] = max(1, int(n))


def f16_train_probe_every() -> int:
    return _f16_train_probe_every[0]


F16_ACT_LIMIT = 65504.0 / 16.0          # largest |activation| the split-fp16 kernel represents (fp16 max / its 2^4 pre-scale)
F16_PREFLIGHT_MARGIN = 4.0               # the probe refuses a model whose sampled activations come within 4x of that limit


def f16_preflight(model, ro, rd, z, rd_view, expr, latent, near, far, max_rays: int = 256, max_samples: int = 8) -> float:
    """Range probe for the split-fp16 kernel: evaluate the EXACT-f32 training forward (which writes every layer's activations)
    on a strided sample of the chunk's points and return the largest hidden |activation|.  ~2k points: microseconds of GPU
    time and one host read-back; called once per frame and model by run_one_iter_of_nerf under "f16x3"."""
    n_rays, n_s = z.shape
    rs = max(1, n_rays // max_rays)
    ss = max(1, n_s // max_samples)
    ro_s, rd_s = ro[::rs].contiguous(), rd[::rs].contiguous()
    z_s = z[::rs, ::ss].contiguous()
    rv_s = None if rd_view is None else rd_view[::rs].contiguous()
    hw = model.hip_weights()
    packed = hw.get()
    cond = paper_condition(packed, expr, latent, near, far)
    _, (saved,) = paper_mlp_fwd_train(packed, cond, ro_s, rd_s, z_s, rv_s)
    n = z_s.numel()
    return float(saved[64 * n:2240 * n].abs().max().item())          # sections S_H0 .. S_D2 (csrc/nf_mlp_layout.h): every hidden layer output


def check_f16_range(*models, sync_ranks: bool = False) -> None:
    """Raise if the split-fp16 kernels of any of `models` flagged a non-finite output since their weights were last packed
    (one 4-byte read-back per model; called once per rendered frame, never inside a ray chunk).
    sync_ranks (data-parallel training): the flag is MAX-reduced over the process group first, so that every rank raises on the
    same iteration -- a rank that raised alone would leave the others waiting in the next gradient all-reduce."""
    flags = [m.hip_weights().f16_range_flag() for m in models if m is not None and hasattr(m, "hip_weights")]
    flags = [f for f in flags if f is not None]
    if not flags:
        return
    total = torch.stack(flags).sum().to(torch.int32).reshape(1)
    if sync_ranks:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(total, op=dist.ReduceOp.MAX)
    if int(total.item()) != 0:
        raise RuntimeError(f'nerf.set_mlp_precision("{_mlp_precision}"): an activation left the fp16 range (|x| >= 4094) and a density output is '
                           'not finite -- render this model with "f32" or "bf16x3"')


def paper_mlp_fwd_train(packed, cond, ro, rd, z, rd_view=None, packed_b=None, packed_h=None):
    """Training forward: returns (raw, (saved,)) where `saved` holds every layer output for the backward.
    packed_b (split-bf16 stream) or packed_h (split-fp16 stream) given -> the forward runs on that split kernel and `saved`
    additionally carries its ReLU bit masks; the matching backward is paper_mlp_bwd(..., split=True / "f16")."""
    dev = H.require_device(packed, cond, ro, rd, z, rd_view)
    n_rays, n_samples = z.shape
    lib = H.lib()
    raw = torch.empty((n_rays, n_samples, 4), dtype=torch.float32, device=dev)
    saved = torch.empty(lib.nf_paper_saved_floats(n_rays * n_samples), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if packed_h is not None:
            H.check(lib.nf_paper_mlp_fwd_train_f16(H.ptr(packed_h), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view), H.ptr(z),
                                                   n_rays, n_samples, H.ptr(raw), H.ptr(saved), H.stream_ptr(dev)),
                    "nf_paper_mlp_fwd_train_f16")
        elif packed_b is not None:
            H.check(lib.nf_paper_mlp_fwd_train_bf16(H.ptr(packed_b), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view), H.ptr(z),
                                                    n_rays, n_samples, H.ptr(raw), H.ptr(saved), H.stream_ptr(dev)),
                    "nf_paper_mlp_fwd_train_bf16")
        else:
            H.check(lib.nf_paper_mlp_fwd_train(H.ptr(packed), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view), H.ptr(z), n_rays,
                                               n_samples, H.ptr(raw), H.ptr(saved), H.stream_ptr(dev)), "nf_paper_mlp_fwd_train")
    return raw, (saved,)


def split_saved_to_f32(saved_t: torch.Tensor, n_points: int, f16: bool, family: str = "paper") -> torch.Tensor:
    """The activations a SPLIT training forward saved (fragment streams of (hi, lo) pairs, sections padded to 32 points) converted to
    the exact-f32 training layout (sections [n_points][width] f32 rows at nfl::S_* x n_points): for tests and for the exact-f32
    weight-gradient GEMMs on a split forward (paper_mlp_bwd(..., exact_dw=True))."""
    dev = H.require_device(saved_t)
    lib = H.lib()
    out = torch.empty((lib.nf_lcode_saved_floats if family == "lcode" else lib.nf_paper_saved_floats)(n_points), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        H.check(lib.nf_split_saved_to_f32(1 if family == "lcode" else 0, H.ptr(saved_t), n_points, int(bool(f16)), H.ptr(out), H.stream_ptr(dev)),
                "nf_split_saved_to_f32")
    return out


def paper_mlp_bwd(model, packed, cond, z, d_raw, saved, split=False, exact_dw=False):
    """d_raw (n_rays, n_samples, 4) -> ([26 parameter gradients in state_dict order], d_latent (32)).
    layers_dir.3.{weight,bias} get None, as autograd gives the reference (Quirk Q3).  split=True runs the dX chain on the
    split-bf16 kernel (requires `saved` from the split-bf16 training forward) and, unless exact_dw, the dW GEMMs too."""
    (saved_t,) = saved
    d_raw = _c(d_raw)
    dev = H.require_device(packed, cond, saved_t, d_raw)
    lib = H.lib()
    n_rays, n_samples = z.shape
    with torch.cuda.device(dev):             # the slice plan behind the size depends on the CURRENT device's CU count (nf_mlp_dw.h): ask on `dev`
        ws_floats = lib.nf_paper_bwd_workspace_floats(n_rays * n_samples)
    ws = torch.empty(ws_floats, dtype=torch.float32, device=dev)
    flat = torch.empty(lib.nf_paper_grad_floats(), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if split == "f16":
            packed_ht = model.hip_weights().get_f16_t()
            H.check(lib.nf_paper_mlp_bwd_f16(H.ptr(packed), H.ptr(packed_ht), H.ptr(cond), H.ptr(saved_t), H.ptr(d_raw), n_rays,
                                             n_samples, H.ptr(ws), ws_floats, H.ptr(flat), H.stream_ptr(dev)), "nf_paper_mlp_bwd_f16")
        elif split:
            packed_bt = model.hip_weights().get_bf16_t()
            saved_f32 = split_saved_to_f32(saved_t, n_rays * n_samples, False) if exact_dw else None      # the exact GEMMs read f32 rows
            H.check(lib.nf_paper_mlp_bwd_bf16(H.ptr(packed), H.ptr(packed_bt), H.ptr(cond), H.ptr(saved_t), H.ptr(d_raw), n_rays,
                                              n_samples, H.ptr(ws), ws_floats, H.ptr(flat), int(bool(exact_dw)), H.ptr(saved_f32),
                                              H.stream_ptr(dev)),
                    "nf_paper_mlp_bwd_bf16")
        else:
            packed_t = model.hip_weights().get_t()
            H.check(lib.nf_paper_mlp_bwd(H.ptr(packed), H.ptr(packed_t), H.ptr(cond), H.ptr(saved_t), H.ptr(d_raw), n_rays,
                                         n_samples, H.ptr(ws), ws_floats, H.ptr(flat), H.stream_ptr(dev)), "nf_paper_mlp_bwd")
    params = model.hip_param_list()
    grads, off = [], 0
    for i, p in enumerate(params):
        n = p.numel()
        grads.append(None if i in (22, 23) else flat[off:off + n].view(p.shape))
        off += n
    return grads, flat[off:off + 32]


# ---------------------------------------------------------------------------------------- K5
def volume_render_fwd(raw, z, rd, noise=None, bg=None, white_background=False):
    dev = H.require_device(raw, z, rd, noise, bg)
    n_rays, n_samples = z.shape
    rgb = torch.empty((n_rays, 3), dtype=torch.float32, device=dev)
    disp = torch.empty((n_rays,), dtype=torch.float32, device=dev)
    acc = torch.empty((n_rays,), dtype=torch.float32, device=dev)
    w = torch.empty((n_rays, n_samples), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        H.check(H.lib().nf_volume_render_fwd(H.ptr(raw), H.ptr(z), H.ptr(rd), H.ptr(noise), H.ptr(bg), n_rays, n_samples,
                                             1 if white_background else 0, H.ptr(rgb), H.ptr(disp), H.ptr(acc), H.ptr(w),
                                             H.stream_ptr(dev)), "nf_volume_render_fwd")
    return rgb, disp, acc, w


def volume_render_bwd(raw, z, rd, noise, bg, d_rgb, white_background=False):
    d_rgb = _c(d_rgb)
    dev = H.require_device(raw, z, rd, noise, bg, d_rgb)
    n_rays, n_samples = z.shape
    d_raw = torch.empty((n_rays, n_samples, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        H.check(H.lib().nf_volume_render_bwd(H.ptr(raw), H.ptr(z), H.ptr(rd), H.ptr(noise), H.ptr(bg), H.ptr(d_rgb), n_rays,
                                             n_samples, 1 if white_background else 0, H.ptr(d_raw), H.stream_ptr(dev)),
                "nf_volume_render_bwd")
    return d_raw


# ---------------------------------------------------------------------------------------- the trainer's loss (TR:355-387)
class _TrainingLoss(torch.autograd.Function):
    """loss = mse(rgb_coarse, target) + mse(rgb_fine, target) + code_scale * code_weight * ||latent||, forward and backward one launch
    each (nf_train_loss_fwd / nf_train_loss_bwd) instead of the ~20 torch launches of TR:355-387 and their backward nodes."""

    @staticmethod
    def forward(ctx, rgb_c, rgb_f, target, latent, code_weight, code_scale):
        dev = H.require_device(rgb_c, rgb_f, target, latent)
        out = torch.empty((7,), dtype=torch.float32, device=dev)
        n = rgb_c.numel()
        with torch.cuda.device(dev):
            H.check(H.lib().nf_train_loss_fwd(H.ptr(rgb_c), H.ptr(rgb_f), H.ptr(target), n, H.ptr(latent), 0 if latent is None else latent.numel(),
                                              float(code_weight), float(code_scale), H.ptr(out), H.stream_ptr(dev)), "nf_train_loss_fwd")
        ctx.save_for_backward(rgb_c, rgb_f, target, latent, out)
        ctx.consts = (float(code_weight), float(code_scale))
        parts = out.detach()
        ctx.mark_non_differentiable(parts)
        return out[0], parts

    @staticmethod
    def backward(ctx, go, _go_parts):
        rgb_c, rgb_f, target, latent, out = ctx.saved_tensors
        dev = rgb_c.device
        go = go.to(torch.float32).contiguous()
        d_c = torch.empty_like(rgb_c)
        d_f = torch.empty_like(rgb_f) if rgb_f is not None else None
        d_l = torch.empty_like(latent) if latent is not None else None
        with torch.cuda.device(dev):
            H.check(H.lib().nf_train_loss_bwd(H.ptr(rgb_c), H.ptr(rgb_f), H.ptr(target), rgb_c.numel(), H.ptr(latent),
                                              0 if latent is None else latent.numel(), ctx.consts[0], ctx.consts[1], H.ptr(out), H.ptr(go),
                                              H.ptr(d_c), H.ptr(d_f), H.ptr(d_l), H.stream_ptr(dev)), "nf_train_loss_bwd")
        return d_c, d_f, None, d_l, None, None


def training_loss(rgb_coarse, rgb_fine, target, latent=None, code_weight: float = 0.0005, code_scale: float = 10.0):
    """The trainer's loss (TR:355-387) fused: returns (loss, parts) with parts = [loss, coarse mse, fine mse, code loss, coarse + fine,
    psnr of coarse + fine, ||latent||] (detached, for logging); `loss` is differentiable w.r.t. rgb_coarse, rgb_fine and latent.
    rgb_fine / latent may be None.  Colour maps and target: the same shape, float32 (made contiguous if they are not)."""
    for t in (rgb_coarse, rgb_fine, target, latent):                     # before _c(): a float64 map must not be down-cast silently
        if t is not None and t.dtype != torch.float32:
            raise TypeError("training_loss: float32 tensors only")
    rgb_c, tgt = _c(rgb_coarse), _c(target)
    rgb_f = _c(rgb_fine) if rgb_fine is not None else None
    lat = _c(latent) if latent is not None else None
    if tgt.shape != rgb_c.shape or (rgb_f is not None and rgb_f.shape != rgb_c.shape):
        raise ValueError(f"training_loss: colour maps {tuple(rgb_c.shape)} / {None if rgb_f is None else tuple(rgb_f.shape)} and target "
                         f"{tuple(tgt.shape)} must have one shape")
    return _TrainingLoss.apply(rgb_c, rgb_f, tgt, lat, code_weight, code_scale)


# ---------------------------------------------------------------------------------------- K6 / K7
def _u_arg(u: Optional[torch.Tensor], n_rays: int, n_out: int, device):
    if u is None:                                     # det mode (H:357-362): linspace(0,1,n) broadcast over rays
        return linspace01(n_out, device), 0
    u = _c(u)
    H.require_device(u)
    assert tuple(u.shape) == (n_rays, n_out), (tuple(u.shape), (n_rays, n_out))
    return u, n_out


def sample_pdf(bins, weights, n_out: int, u: Optional[torch.Tensor] = None, want_table: bool = False):
    """sample_pdf_2 (H:344-387).  want_table: also return (inds int32 (R,n_out) = searchsorted(cdf, u, right=True), cdf (R,n_bins))."""
    bins, weights = _c(bins), _c(weights)
    dev = H.require_device(bins, weights)
    n_rays, n_bins = bins.shape
    assert weights.shape == (n_rays, n_bins - 1)
    u_t, stride = _u_arg(u, n_rays, n_out, dev)
    out = torch.empty((n_rays, n_out), dtype=torch.float32, device=dev)
    inds = torch.empty((n_rays, n_out), dtype=torch.int32, device=dev) if want_table else None
    cdf = torch.empty((n_rays, n_bins), dtype=torch.float32, device=dev) if want_table else None
    with torch.cuda.device(dev):
        H.check(H.lib().nf_sample_pdf_ex(H.ptr(bins), H.ptr(weights), H.ptr(u_t), stride, n_rays, n_bins, n_out, H.ptr(out),
                                         H.ptr(inds), H.ptr(cdf), H.stream_ptr(dev)), "nf_sample_pdf_ex")
    return (out, inds, cdf) if want_table else out


def resample_merge(z_coarse, w_coarse, n_fine: int, u: Optional[torch.Tensor] = None, want_samples: bool = False):
    dev = H.require_device(z_coarse, w_coarse)
    n_rays, n_coarse = z_coarse.shape
    u_t, stride = _u_arg(u, n_rays, n_fine, dev)
    z_fine = torch.empty((n_rays, n_coarse + n_fine), dtype=torch.float32, device=dev)
    z_s = torch.empty((n_rays, n_fine), dtype=torch.float32, device=dev) if want_samples else None
    with torch.cuda.device(dev):
        H.check(H.lib().nf_resample_merge(H.ptr(z_coarse), H.ptr(w_coarse), H.ptr(u_t), stride, n_rays, n_coarse, n_fine,
                                          H.ptr(z_s), H.ptr(z_fine), H.stream_ptr(dev)), "nf_resample_merge")
    return (z_fine, z_s) if want_samples else z_fine


def sort_rows(x: torch.Tensor) -> torch.Tensor:
    x = _c(x)
    dev = H.require_device(x)
    n_cols = x.shape[-1]
    rows = x.numel() // n_cols
    out = torch.empty_like(x)
    with torch.cuda.device(dev):
        H.check(H.lib().nf_sort_rows(H.ptr(x), rows, n_cols, H.ptr(out), H.stream_ptr(dev)), "nf_sort_rows")
    return out


# ---------------------------------------------------------------------------------------- eval post-processing
def eval_postprocess(rgb, depthmap=None, weights=None, intrinsics=None, want_normals=True):
    """(H,W,3) float image -> uint8 image (clamp, x255, truncate: EV:184-190) and, from the disparity/"depth" map the eval
    script feeds torch_normal_map (EV:84-119), the cleaned uint8 normal map (H-1, W-1, 3).  Device tensors in and out."""
    rgb = _c(rgb)
    depthmap = _c(depthmap) if depthmap is not None else None
    weights = _c(weights) if weights is not None else None
    dev = H.require_device(rgb, depthmap, weights)
    h, w = rgb.shape[0], rgb.shape[1]
    rgb_u8 = torch.empty((h, w, 3), dtype=torch.uint8, device=dev)
    normals = None
    fx = fy = 1.0
    cx = cy = 0.0
    if want_normals and depthmap is not None:
        import numpy as _np
        a = _np.asarray(intrinsics, dtype=_np.float64).reshape(-1)
        fx, fy = float(_np.float32(a[0])), float(_np.float32(a[1]))
        cx, cy = float(_np.float32(a[2] * w)), float(_np.float32(a[3] * h))
        normals = torch.empty((h - 1, w - 1, 3), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        H.check(H.lib().nf_eval_postprocess(H.ptr(rgb), H.ptr(depthmap), H.ptr(weights), h, w, fx, fy, cx, cy, H.ptr(rgb_u8),
                                            H.ptr(normals), H.stream_ptr(dev)), "nf_eval_postprocess")
    return rgb_u8, normals
