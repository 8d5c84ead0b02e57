"""MI355X-native counterparts of the reference's nerf/nerf_helpers.py (same names, same signatures).

Every numerical function here is a thin call into libnerface_hip.so; tensors must be on a ROCm device.
Reference citations: H = nerf/nerf_helpers.py of gafniguy/4D-Facial-Avatars.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from . import ops


def img2mse(img_src, img_tgt):
    """H:10 -- loss glue of the caller scripts (torch op on whatever device the images are on)."""
    return torch.nn.functional.mse_loss(img_src, img_tgt)


def mse2psnr(mse):
    """H:14-18."""
    if mse == 0:
        mse = 1e-5
    return -10.0 * math.log10(mse)


def get_minibatches(inputs: torch.Tensor, chunksize: Optional[int] = 1024 * 8):
    """H:21-26: list of row chunks (views)."""
    return [inputs[i: i + chunksize] for i in range(0, inputs.shape[0], chunksize)]


def meshgrid_xy(tensor1: torch.Tensor, tensor2: torch.Tensor):
    """H:29-41: np.meshgrid(..., indexing='xy') for two 1-D tensors (index glue used by the trainer)."""
    ii, jj = torch.meshgrid(tensor1, tensor2, indexing="ij")
    return ii.transpose(-1, -2), jj.transpose(-1, -2)


def cumprod_exclusive(tensor: torch.Tensor) -> torch.Tensor:
    """H:44-65.  Kept for API compatibility (tiny_nerf.py imports it); the hot path computes the exclusive
    transmittance product inside the volume-integrator kernel (wavefront scan)."""
    cp = torch.cumprod(tensor, -1)
    cp = torch.roll(cp, 1, -1)
    cp[..., 0] = 1.0
    return cp


def _intrinsics4(intrinsics):
    """[fx, fy, cx_rel, cy_rel]; a scalar focal length falls back to [f, f, .5, .5] (H:109-110)."""
    if torch.is_tensor(intrinsics):
        intrinsics = intrinsics.detach().cpu().numpy()
    a = np.atleast_1d(np.asarray(intrinsics, dtype=np.float64)).reshape(-1)
    if a.shape[0] < 4:
        return float(a[0]), float(a[0]), 0.5, 0.5
    return float(a[0]), float(a[1]), float(a[2]), float(a[3])


def get_ray_bundle(height: int, width: int, intrinsics, tform_cam2world: torch.Tensor, center=[0.5, 0.5]):
    """H:68-123 -> kernel K1 (nf_ray_bundle).  Returns (ray_origins, ray_directions), each (H, W, 3)."""
    fx, fy, cx, cy = _intrinsics4(intrinsics)
    return ops.ray_bundle(int(height), int(width), fx, fy, cx, cy, tform_cam2world)


def choose_rays(probs: torch.Tensor, n: int, check: bool = False) -> torch.Tensor:
    """The trainer's ray selection (TR:320-322: np.random.choice(H * W, size=n, replace=False, p=probs) on the host) on the device:
    n distinct pixel indices drawn without replacement with probabilities proportional to `probs` (nf_weighted_choice).  Not in the
    reference (the launchers' replacement); the random numbers come from torch's device generator."""
    return ops.weighted_choice(probs, int(n), check=check)


def get_ray_batch(height: int, width: int, intrinsics, tform_cam2world: torch.Tensor, select_inds: torch.Tensor, target_img=None,
                  background=None, check: bool = False):
    """Training-batch form of get_ray_bundle (not in the reference: the launcher's replacement for TR:302 + TR:325-330):
    rays, target pixels and background prior of the pixels select_inds -- (n, 2) = {row, col}, or (n,) flat indices as choose_rays
    returns them -- only, one kernel (nf_ray_batch).
    Rays are bit-identical to get_ray_bundle(...)[select_inds[:, 0], select_inds[:, 1]]."""
    fx, fy, cx, cy = _intrinsics4(intrinsics)
    return ops.ray_batch(int(height), int(width), fx, fy, cx, cy, tform_cam2world, select_inds, target_img, background, check)


class PositionalEncoder:
    """Callable returned by get_embedding_function.  It is tagged with its parameters so that
    run_one_iter_of_nerf can recognise it and run the encoding inside the fused MLP kernel instead of
    materialising the (rays*samples) x 87 tensor the reference builds (T:11-18)."""

    def __init__(self, num_encoding_functions, include_input, log_sampling):
        self.num_encoding_functions = int(num_encoding_functions)
        self.include_input = bool(include_input)
        self.log_sampling = bool(log_sampling)

    def __call__(self, x):
        return positional_encoding(x, self.num_encoding_functions, self.include_input, self.log_sampling)


def positional_encoding(tensor, num_encoding_functions=6, include_input=True, log_sampling=True) -> torch.Tensor:
    """H:195-239 -> kernel K3 (nf_posenc)."""
    if not log_sampling:
        raise NotImplementedError("log_sampling=False is not used by any NeRFace config and is not built")
    if num_encoding_functions == 0 and include_input:
        return tensor
    return ops.posenc(tensor, int(num_encoding_functions), bool(include_input))


def get_embedding_function(num_encoding_functions=6, include_input=True, log_sampling=True):
    """H:242-249."""
    return PositionalEncoder(num_encoding_functions, include_input, log_sampling)


def sample_pdf_2(bins, weights, num_samples, det=False):
    """H:344-387 -> kernel K6 (nf_sample_pdf).  det=False draws u with torch.rand on the tensors' device,
    exactly as the reference does (H:363-367)."""
    u = None
    if not det:
        u = torch.rand(list(bins.shape[:-1]) + [num_samples], dtype=weights.dtype, device=weights.device)
    return ops.sample_pdf(bins, weights, int(num_samples), u)


sample_pdf = sample_pdf_2


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """H:252-281.  Every shipped NeRFace config sets dataset.no_ndc: True; the NDC warp is out of scope."""
    raise NotImplementedError("NDC rays are not part of the NeRFace path (all configs use no_ndc: True)")


def dump_rays(origins, points, radiance_field):
    """H:389-...: PLY debugging dump; host-side tooling outside the hot path."""
    raise NotImplementedError("dump_rays is a debugging helper of the reference and is not provided")
