"""MI355X-native counterpart of the reference's tiny_nerf.py (BASELINE config 1): same function names and signatures
(`compute_query_points_from_rays`, `render_volume_density`, `run_one_iter_of_tinynerf`, `VeryTinyNerfModel`), inference
on libnerface_hip.so.  Reference citations: TN = tiny_nerf.py of gafniguy/4D-Facial-Avatars.

The forward pass (what TN:111-159 computes) is one fused HIP kernel for query points + positional encoding + the
3-layer MLP (nf_tiny_mlp_fwd) and one for the compositing (nf_render_volume_density).  Training tiny_nerf (autograd through
these kernels) is not provided: the product's training path is the NeRFace trainer (nerf.run_one_iter_of_nerf).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from nerf import _hip as H
from nerf import get_minibatches, get_ray_bundle, positional_encoding  # noqa: F401  (same imports as the reference script)
from nerf.ops import _c


def _depths(ray_origins, near_thresh, far_thresh, num_samples, randomize):
    """TN:46-57: linspace(near, far, S) (+ rand * (far - near) / S per ray and sample)."""
    depth_values = torch.linspace(near_thresh, far_thresh, num_samples).to(ray_origins)
    if randomize is True:
        noise_shape = list(ray_origins.shape[:-1]) + [num_samples]
        depth_values = depth_values + torch.rand(noise_shape).to(ray_origins) * (far_thresh - near_thresh) / num_samples
    return depth_values


def compute_query_points_from_rays(ray_origins, ray_directions, near_thresh, far_thresh, num_samples, randomize: Optional[bool] = True):
    """TN:12-65 (API compatibility; run_one_iter_of_tinynerf never materialises the points)."""
    depth_values = _depths(ray_origins, near_thresh, far_thresh, num_samples, randomize)
    query_points = ray_origins[..., None, :] + ray_directions[..., None, :] * depth_values[..., :, None]
    return query_points, depth_values


def render_volume_density(radiance_field, ray_origins, depth_values):
    """TN:68-107 -> (rgb_map, depth_map, acc_map) via nf_render_volume_density."""
    raw = _c(radiance_field)
    S = raw.shape[-2]
    lead = raw.shape[:-2]
    depth = _c(depth_values.expand(*lead, S)) if depth_values.dim() > 1 else _c(depth_values.expand(*lead, S))
    dev = H.require_device(raw, depth)
    n = raw.numel() // (4 * S)
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    dmap = torch.empty((n,), dtype=torch.float32, device=dev)
    acc = torch.empty((n,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        H.check(H.lib().nf_render_volume_density(H.ptr(raw), H.ptr(depth), n, S, H.ptr(rgb), H.ptr(dmap), H.ptr(acc),
                                                 H.stream_ptr(dev)), "nf_render_volume_density")
    return rgb.reshape(*lead, 3), dmap.reshape(lead), acc.reshape(lead)


class VeryTinyNerfModel(torch.nn.Module):
    """TN:162-181: three Linear layers, 3 + 6 n -> filter -> filter -> 4.  The HIP kernel is built for the reference's
    configuration (filter_size=128, num_encoding_functions=10 as at TN:264-276)."""

    def __init__(self, filter_size=128, num_encoding_functions=6):
        super().__init__()
        self.layer1 = torch.nn.Linear(3 + 3 * 2 * num_encoding_functions, filter_size)
        self.layer2 = torch.nn.Linear(filter_size, filter_size)
        self.layer3 = torch.nn.Linear(filter_size, 4)
        self.relu = torch.nn.functional.relu
        self._packed = None
        self._sig = None

    def fused_supported(self):
        return self.layer1.in_features == 63 and self.layer1.out_features == 128

    def hip_packed(self):
        ps = [self.layer1.weight, self.layer1.bias, self.layer2.weight, self.layer2.bias, self.layer3.weight, self.layer3.bias]
        sig = tuple((int(p.data_ptr()), int(p._version)) for p in ps)
        if self._packed is None or sig != self._sig:
            dev = H.require_device(*[p.detach() for p in ps])
            lib = H.lib()
            self._packed = torch.empty(lib.nf_tiny_packed_floats(), dtype=torch.float32, device=dev)
            arr = (C.c_void_p * 6)(*[int(p.data_ptr()) for p in ps])
            with torch.cuda.device(dev):
                H.check(lib.nf_tiny_pack(arr, H.ptr(self._packed), H.stream_ptr(dev)), "nf_tiny_pack")
            self._sig = sig
        return self._packed

    def forward(self, x):
        raise NotImplementedError("VeryTinyNerfModel is evaluated inside the fused kernel: call run_one_iter_of_tinynerf(...)")


def run_one_iter_of_tinynerf(height, width, focal_length, tform_cam2world, near_thresh, far_thresh, depth_samples_per_ray,
                             encoding_function, get_minibatches_function, chunksize, model, encoding_function_args):
    """TN:111-159 (same signature; `encoding_function`, `get_minibatches_function` and `chunksize` are accepted for
    compatibility -- encoding and chunking happen inside the fused kernel).  Returns rgb_predicted (H, W, 3)."""
    if not isinstance(model, VeryTinyNerfModel) or not model.fused_supported() or int(encoding_function_args) != 10:
        raise NotImplementedError("the fused tiny kernel is built for VeryTinyNerfModel(128, num_encoding_functions=10)")
    if torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()):
        raise NotImplementedError("tiny_nerf training is not provided by the MI355X build; wrap inference in torch.no_grad()")
    ray_origins, ray_directions = get_ray_bundle(height, width, focal_length, tform_cam2world)
    depth_values = _depths(ray_origins, near_thresh, far_thresh, depth_samples_per_ray, True)       # default randomize=True
    dev = ray_origins.device
    ro, rd = ray_origins.reshape(-1, 3), ray_directions.reshape(-1, 3)
    depth = _c(depth_values.reshape(-1, depth_samples_per_ray))
    n = ro.shape[0]
    raw = torch.empty((n, depth_samples_per_ray, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        H.check(H.lib().nf_tiny_mlp_fwd(H.ptr(model.hip_packed()), H.ptr(_c(ro)), H.ptr(_c(rd)), H.ptr(depth), 1, n,
                                        depth_samples_per_ray, H.ptr(raw), H.stream_ptr(dev)), "nf_tiny_mlp_fwd")
    rgb, _, _ = render_volume_density(raw.reshape(height, width, depth_samples_per_ray, 4), ray_origins, depth_values)
    return rgb
