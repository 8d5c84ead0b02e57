"""MI355X-native volume_render_radiance_field (reference nerf/volume_rendering_utils.py:7-75)."""
from __future__ import annotations

import torch

from . import ops


class _VolumeRender(torch.autograd.Function):
    """K5 forward/backward.  Differentiable w.r.t. `radiance_field` through rgb_map only (that is the only
    path the trainer's loss uses, train_transformed_rays.py:355-387); the other outputs are marked
    non-differentiable."""

    @staticmethod
    def forward(ctx, raw, z, rd, noise, bg, white_background):
        rgb, disp, acc, w = ops.volume_render_fwd(raw, z, rd, noise, bg, white_background)
        ctx.save_for_backward(raw, z, rd, noise, bg)
        ctx.white_background = white_background
        ctx.mark_non_differentiable(disp, acc, w)
        return rgb, disp, acc, w

    @staticmethod
    def backward(ctx, d_rgb, d_disp, d_acc, d_w):
        raw, z, rd, noise, bg = ctx.saved_tensors
        d_raw = ops.volume_render_bwd(raw, z, rd, noise, bg, d_rgb.contiguous(), ctx.white_background)
        return d_raw, None, None, None, None, None


def volume_render_radiance_field(radiance_field, depth_values, ray_directions, radiance_field_noise_std=0.0,
                                 white_background=False, background_prior=None):
    """Same signature and return tuple as the reference: (rgb_map, disp_map, acc_map, weights, None).

    Like the reference, the caller is expected to have overwritten radiance_field[:, -1, :3] with the
    background when `background_prior` is given (train_utils.py:95-96); the kernel reads the colour of the
    last sample from `background_prior` directly, which is the same value."""
    raw = ops._c(radiance_field)
    z = ops._c(depth_values)
    rd = ops._c(ray_directions)
    bg = ops._c(background_prior) if background_prior is not None else None
    noise = None
    if radiance_field_noise_std > 0.0:
        noise = torch.randn(raw[..., 3].shape, dtype=raw.dtype, device=raw.device) * radiance_field_noise_std
    lead = z.shape[:-1]
    S = z.shape[-1]
    rgb, disp, acc, w = _VolumeRender.apply(raw.reshape(-1, S, 4), z.reshape(-1, S), rd.reshape(-1, 3),
                                            None if noise is None else noise.reshape(-1, S),
                                            None if bg is None else bg.reshape(-1, 3), bool(white_background))
    return rgb.reshape(*lead, 3), disp.reshape(lead), acc.reshape(lead), w.reshape(*lead, S), None
