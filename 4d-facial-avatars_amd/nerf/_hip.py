"""ctypes binding of libnerface_hip.so (C ABI declared in include/nerface_hip.h).

The library is the product: there is no fallback.  If it has not been built, or a tensor is not on a
ROCm device, the callers in this package raise -- they never route to a CPU/PyTorch implementation.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch  # noqa: F401  (must be imported first: it loads the HIP runtime the library binds to)

_LIB_PATH = os.environ.get("NERFACE_HIP_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib",
                                                             "libnerface_hip.so")
_lib = None
_lock = threading.Lock()

NF_PAPER_NUM_PARAMS = 26

_P, _I, _L, _F, _Z = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
_PROTOTYPES = {
    "nf_abi_version": (C.c_int, []),
    "nf_error_string": (C.c_char_p, [_I]),
    "nf_build_info": (C.c_char_p, []),
    "nf_ray_bundle": (C.c_int, [_I, _I, _F, _F, _F, _F, _P, _I, _P, _P, _P]),
    "nf_weighted_choice_workspace_bytes": (_Z, []),
    "nf_weighted_choice": (C.c_int, [_P, _P, _L, _I, _P, _P, _Z, _P]),
    "nf_ray_batch": (C.c_int, [_I, _I, _F, _F, _F, _F, _P, _I, _P, _I, _L, _P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "nf_sample_coarse": (C.c_int, [_L, _I, _F, _F, _P, _P, _P, _P]),
    "nf_sample_coarse_ex": (C.c_int, [_L, _I, _F, _F, _P, _P, _I, _P, _P]),
    "nf_posenc": (C.c_int, [_P, _L, _I, _I, _I, _P, _P]),
    "nf_paper_packed_floats": (_Z, []),
    "nf_paper_cond_floats": (_Z, []),
    "nf_paper_gather_table": (C.c_int, [_P, _Z]),
    "nf_paper_pack": (C.c_int, [_P, _P, _P]),
    "nf_paper_condition": (C.c_int, [_P, _P, _P, _F, _F, _P, _P]),
    "nf_paper_mlp_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "nf_paper_forward_encoded": (C.c_int, [_P, _P, _P, _P, _L, _P, _P, _P]),
    "nf_paper_packed_bf16_bytes": (_Z, []),
    "nf_paper_pack_bf16": (C.c_int, [_P, _P, _P]),
    "nf_paper_mlp_fwd_bf16": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "nf_paper_packed_f16_bytes": (_Z, []),
    "nf_paper_f16_flag_offset": (_Z, []),
    "nf_paper_pack_f16": (C.c_int, [_P, _P, _P]),
    "nf_paper_mlp_fwd_f16": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "nf_paper_mlp_fwd_f16x2": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "nf_paper_mlp_fwd_train_f16": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "nf_paper_packed_bwd_f16_bytes": (_Z, []),
    "nf_paper_pack_bwd_f16": (C.c_int, [_P, _P, _P]),
    "nf_paper_mlp_bwd_f16": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _P, _Z, _P, _P]),
    "nf_paper_mlp_fwd_train_bf16": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "nf_paper_saved_floats": (_Z, [_L]),
    "nf_paper_mlp_fwd_train": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "nf_paper_packed_bwd_floats": (_Z, []),
    "nf_paper_pack_bwd": (C.c_int, [_P, _P, _P]),
    "nf_paper_packed_bwd_bf16_bytes": (_Z, []),
    "nf_paper_pack_bwd_bf16": (C.c_int, [_P, _P, _P]),
    "nf_paper_mlp_bwd_bf16": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _P, _Z, _P, _I, _P, _P]),
    "nf_split_saved_to_f32": (C.c_int, [_I, _P, _L, _I, _P, _P]),
    "nf_paper_grad_floats": (_Z, []),
    "nf_paper_bwd_workspace_floats": (_Z, [_L]),
    "nf_paper_mlp_bwd": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _P, _Z, _P, _P]),
    "nf_paper_mlp_bwd_stage_ms": (C.c_int, [_P, _P, _I, _P, _P, _P, _L, _I, _P, _Z, _P, _P, _P]),
    "nf_lcode_packed_floats": (_Z, []),
    "nf_lcode_cond_floats": (_Z, []),
    "nf_lcode_pack": (C.c_int, [_P, _P, _P]),
    "nf_lcode_condition": (C.c_int, [_P, _P, _P, _F, _F, _P, _P]),
    "nf_lcode_mlp_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "nf_lcode_packed_bf16_bytes": (_Z, []),
    "nf_lcode_pack_bf16": (C.c_int, [_P, _P, _P]),
    "nf_lcode_mlp_fwd_bf16": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "nf_lcode_packed_f16_bytes": (_Z, []),
    "nf_lcode_f16_flag_offset": (_Z, []),
    "nf_lcode_pack_f16": (C.c_int, [_P, _P, _P]),
    "nf_lcode_mlp_fwd_f16": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "nf_lcode_mlp_fwd_f16x2": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P]),
    "nf_lcode_mlp_fwd_train_f16": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "nf_lcode_packed_bwd_f16_bytes": (_Z, []),
    "nf_lcode_pack_bwd_f16": (C.c_int, [_P, _P, _P]),
    "nf_lcode_mlp_bwd_f16": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _P, _Z, _P, _P]),
    "nf_lcode_mlp_fwd_train_bf16": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "nf_lcode_packed_bwd_bf16_bytes": (_Z, []),
    "nf_lcode_pack_bwd_bf16": (C.c_int, [_P, _P, _P]),
    "nf_lcode_mlp_bwd_bf16": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _P, _Z, _P, _P]),
    "nf_selftest_dw_tables_f32": (C.c_int, []),
    "nf_selftest_dw_tables_lcode_f32": (C.c_int, []),
    "nf_selftest_dw_tables_bf16": (C.c_int, []),
    "nf_paper_stream_table_bf16": (C.c_long, [_P, _Z]),
    "nf_paper_stream_table_bwd_bf16": (C.c_long, [_P, _Z]),
    "nf_lcode_stream_table_bf16": (C.c_long, [_P, _Z]),
    "nf_lcode_stream_table_bwd_bf16": (C.c_long, [_P, _Z]),
    "nf_lcode_forward_encoded": (C.c_int, [_P, _P, _P, _P, _L, _P, _P, _P]),
    "nf_lcode_saved_floats": (_Z, [_L]),
    "nf_lcode_mlp_fwd_train": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "nf_lcode_packed_bwd_floats": (_Z, []),
    "nf_lcode_pack_bwd": (C.c_int, [_P, _P, _P]),
    "nf_lcode_grad_floats": (_Z, []),
    "nf_lcode_bwd_workspace_floats": (_Z, [_L]),
    "nf_lcode_mlp_bwd": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _P, _Z, _P, _P]),
    "nf_eval_postprocess": (C.c_int, [_P, _P, _P, _I, _I, _F, _F, _F, _F, _P, _P, _P]),
    "nf_tiny_packed_floats": (_Z, []),
    "nf_tiny_pack": (C.c_int, [_P, _P, _P]),
    "nf_tiny_mlp_fwd": (C.c_int, [_P, _P, _P, _P, _I, _L, _I, _P, _P]),
    "nf_render_volume_density": (C.c_int, [_P, _P, _L, _I, _P, _P, _P, _P]),
    "nf_tiny_saved_floats": (_Z, [_L]),
    "nf_tiny_mlp_fwd_train": (C.c_int, [_P, _P, _P, _P, _I, _L, _I, _P, _P, _P]),
    "nf_render_volume_density_bwd": (C.c_int, [_P, _P, _P, _L, _I, _P, _P]),
    "nf_tiny_packed_bwd_floats": (_Z, []),
    "nf_tiny_pack_bwd": (C.c_int, [_P, _P, _P]),
    "nf_tiny_grad_floats": (_Z, []),
    "nf_tiny_bwd_workspace_floats": (_Z, [_L]),
    "nf_tiny_mlp_bwd": (C.c_int, [_P, _P, _P, _L, _I, _P, _Z, _P, _P]),
    "nf_selftest_dw_tables_tiny": (C.c_int, []),
    "nf_flex_packed_floats": (_Z, [_I]),
    "nf_flex_pack": (C.c_int, [_I, _P, _P, _P]),
    "nf_flex_mlp_fwd": (C.c_int, [_I, _P, _P, _P, _P, _I, _L, _I, _P, _P]),
    "nf_flex_saved_floats": (_Z, [_I, _L]),
    "nf_flex_mlp_fwd_train": (C.c_int, [_I, _P, _P, _P, _P, _I, _L, _I, _P, _P, _P]),
    "nf_flex_packed_bwd_floats": (_Z, [_I]),
    "nf_flex_pack_bwd": (C.c_int, [_I, _P, _P, _P]),
    "nf_flex_grad_floats": (_Z, [_I]),
    "nf_flex_bwd_workspace_floats": (_Z, [_I, _L]),
    "nf_flex_mlp_bwd": (C.c_int, [_I, _P, _P, _P, _L, _I, _P, _Z, _P, _P]),
    "nf_selftest_dw_tables_flex": (C.c_int, [_I]),
    "nf_volume_render_fwd": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P, _P]),
    "nf_volume_render_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P]),
    "nf_sample_pdf": (C.c_int, [_P, _P, _P, _L, _L, _I, _I, _P, _P]),
    "nf_sample_pdf_ex": (C.c_int, [_P, _P, _P, _L, _L, _I, _I, _P, _P, _P, _P]),
    "nf_resample_merge": (C.c_int, [_P, _P, _P, _L, _L, _I, _I, _P, _P, _P]),
    "nf_render_rays_workspace_floats": (_Z, [_L, _I, _I]),
    "nf_render_rays_fwd": (C.c_int, [_P] * 13 + [_L, _P, _P, _L, _I, _I, _F, _F, _I, _P, _Z] + [_P] * 7 + [_P]),
    "nf_render_rays_fwd_f16": (C.c_int, [_P] * 13 + [_L, _P, _P, _L, _I, _I, _F, _F, _I, _P, _Z] + [_P] * 7 + [_P]),
    "nf_render_rays_fwd_f16x2": (C.c_int, [_P] * 13 + [_L, _P, _P, _L, _I, _I, _F, _F, _I, _P, _Z] + [_P] * 7 + [_P]),
    "nf_adam_step": (C.c_int, [_P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _L, _P]),
    "nf_train_loss_fwd": (C.c_int, [_P, _P, _P, _L, _P, _I, _F, _F, _P, _P]),
    "nf_train_loss_bwd": (C.c_int, [_P, _P, _P, _L, _P, _I, _F, _F, _P, _P, _P, _P, _P, _P]),
    "nf_sort_rows": (C.c_int, [_P, _L, _I, _P, _P]),
}
# entry points that later ABI revisions add; absent symbols only fail when called
_OPTIONAL = set()
# the revision of include/nerface_hip.h these prototypes were written for (nf_abi_version() of the library must equal it: a stale
# .so with other signatures would take e.g. a stream pointer as `saved_f32` without any error)
ABI_VERSION = 5


def lib_path() -> str:
    return _LIB_PATH


def lib():
    """dlopen the library (once).  Raises RuntimeError with build instructions when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"libnerface_hip.so not found at {_LIB_PATH}: build it with `python 4d-facial-avatars_amd/build.py` "
                "(or __graft_entry__.build()).  This package has no CPU / PyTorch fallback.")
        handle = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _PROTOTYPES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                if name in _OPTIONAL:
                    continue
                raise RuntimeError(f"libnerface_hip.so does not export {name}; rebuild it")
            fn.restype, fn.argtypes = res, args
        if handle.nf_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libnerface_hip.so at {_LIB_PATH} has ABI revision {handle.nf_abi_version()}, this binding needs {ABI_VERSION}: "
                               "rebuild it (python 4d-facial-avatars_amd/build.py --force)")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().nf_error_string(rc)
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else rc} (code {rc})")


def stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def require_device(*tensors) -> torch.device:
    """All tensors must be fp32, contiguous and on one ROCm device; returns that device."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("nerf (MI355X build): tensors must live on a ROCm device (`device='cuda'`); "
                               "there is no CPU path in this package")
        if t.dtype != torch.float32:
            raise RuntimeError(f"nerf (MI355X build): expected float32 tensors, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError("nerf (MI355X build): internal error, tensor is not contiguous")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError("nerf (MI355X build): tensors are on different devices")
    return dev


def ptr(t) -> int:
    return 0 if t is None else int(t.data_ptr())
