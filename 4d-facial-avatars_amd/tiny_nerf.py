"""MI355X-native counterpart of the reference's tiny_nerf.py (BASELINE config 1): same function names and signatures
(`compute_query_points_from_rays`, `render_volume_density`, `run_one_iter_of_tinynerf`, `VeryTinyNerfModel`), inference
on libnerface_hip.so.  Reference citations: TN = tiny_nerf.py of gafniguy/4D-Facial-Avatars.

The forward pass (what TN:111-159 computes) is one fused HIP kernel for query points + positional encoding + the
3-layer MLP (nf_tiny_mlp_fwd) and one for the compositing (nf_render_volume_density).  With gradients enabled the same
call is differentiable w.r.t. the six model parameters (the reference script is a trainer, TN:282-302): the training
forward saves PE / h1 / h2, and the backward runs nf_render_volume_density_bwd + nf_tiny_mlp_bwd (exact f32).

BASELINE config 1 says "4-layer MLP"; the script's own model has three Linear layers.  Both readings run: `model` may also be the
reference's `nerf.models.FlexibleNeRFModel(num_layers=4, hidden_size=128, num_encoding_fn_xyz=10, use_viewdirs=False)` (M:351-422;
2 .. 5 layers), which the reference's run_one_iter_of_tinynerf accepts as it accepts any module mapping (N, 63) to (N, 4) -- the
nf_flex_* kernels, forward and backward.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from nerf import _hip as H
from nerf import get_minibatches, get_ray_bundle, positional_encoding  # noqa: F401  (same imports as the reference script)
from nerf.models import FlexibleNeRFModel
from nerf.ops import _c, bump_pack_epoch, pack_epoch


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


def _api(model):
    """(entry-point prefix, leading arguments) of the kernels built for `model`: nf_tiny_* for VeryTinyNerfModel(128, 10),
    nf_flex_*(num_layers, ...) for FlexibleNeRFModel(L, 128, num_encoding_fn_xyz=10, use_viewdirs=False)."""
    if isinstance(model, VeryTinyNerfModel) and model.fused_supported():
        return "nf_tiny", ()
    if isinstance(model, FlexibleNeRFModel) and model.fused_supported():
        return "nf_flex", (model.num_layers,)
    raise NotImplementedError("the fused tiny kernels are built for VeryTinyNerfModel(128, num_encoding_functions=10) and "
                              "FlexibleNeRFModel(num_layers=2..5, hidden_size=128, num_encoding_fn_xyz=10, use_viewdirs=False)")


def _fn(model, name):
    """The C entry point `<prefix>_<name>` of the model's kernel family with the family's leading arguments bound."""
    prefix, lead = _api(model)
    f = getattr(H.lib(), f"{prefix}_{name}")
    return lambda *args: f(*lead, *args)


def _image(model, kind):
    """Fragment-ordered weight image of a tiny-path model (kind "pack": forward, "pack_bwd": transposed, for the backward chain),
    rebuilt when a parameter's storage / version counter or the pack epoch moves (fused optimizers do not bump version counters)."""
    ps = model.hip_param_list()
    sig = (pack_epoch(),) + tuple((int(p.data_ptr()), int(p._version)) for p in ps)
    cache = model.__dict__.setdefault("_tiny_images", {})
    hit = cache.get(kind)
    if hit is None or hit[0] != sig:
        dev = H.require_device(*[p.detach() for p in ps])
        size = _fn(model, "packed_floats" if kind == "pack" else "packed_bwd_floats")()
        img = torch.empty(size, dtype=torch.float32, device=dev)
        arr = (C.c_void_p * len(ps))(*[int(p.data_ptr()) for p in ps])
        with torch.cuda.device(dev):
            H.check(_fn(model, kind)(arr, H.ptr(img), H.stream_ptr(dev)), f"{_api(model)[0]}_{kind}")
        cache[kind] = (sig, img)
        hit = cache[kind]
    return hit[1]


class VeryTinyNerfModel(torch.nn.Module):
    """TN:162-181: three Linear layers, 3 + 6 n -> filter -> filter -> 4.  The HIP kernel is built for the reference's
    configuration (filter_size=128, num_encoding_functions=10 as at TN:264-276)."""

    def __init__(self, filter_size=128, num_encoding_functions=6):
        super().__init__()
        self.layer1 = torch.nn.Linear(3 + 3 * 2 * num_encoding_functions, filter_size)
        self.layer2 = torch.nn.Linear(filter_size, filter_size)
        self.layer3 = torch.nn.Linear(filter_size, 4)
        self.relu = torch.nn.functional.relu

    def fused_supported(self):
        return self.layer1.in_features == 63 and self.layer1.out_features == 128

    def hip_param_list(self):
        return [self.layer1.weight, self.layer1.bias, self.layer2.weight, self.layer2.bias, self.layer3.weight, self.layer3.bias]

    def hip_packed(self):
        return _image(self, "pack")

    def hip_packed_t(self):
        """Transposed fragment image of layer2 / layer3 for the backward chain (cached like hip_packed)."""
        return _image(self, "pack_bwd")

    def forward(self, x):
        raise NotImplementedError("VeryTinyNerfModel is evaluated inside the fused kernel: call run_one_iter_of_tinynerf(...)")


class _TinyRender(torch.autograd.Function):
    """rgb (n_rays, 3) = render_volume_density(model(PE(ro + rd * depth))) with gradients for the model's parameters
    (TN:111-159 under autograd; model: VeryTinyNerfModel or FlexibleNeRFModel, `params` = model.hip_param_list()).  Ray origins /
    directions / depths carry no gradient (nothing upstream is learnable)."""

    @staticmethod
    def forward(ctx, model, ro, rd, depth, n_samples, *params):
        dev = ro.device
        n = ro.shape[0]
        lib = H.lib()
        raw = torch.empty((n, n_samples, 4), dtype=torch.float32, device=dev)
        saved = torch.empty(_fn(model, "saved_floats")(n * n_samples), dtype=torch.float32, device=dev)
        rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
        dmap = torch.empty((n,), dtype=torch.float32, device=dev)
        acc = torch.empty((n,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            H.check(_fn(model, "mlp_fwd_train")(H.ptr(_image(model, "pack")), H.ptr(ro), H.ptr(rd), H.ptr(depth), 1, n, n_samples, H.ptr(raw),
                                                H.ptr(saved), H.stream_ptr(dev)), "mlp_fwd_train")
            H.check(lib.nf_render_volume_density(H.ptr(raw), H.ptr(depth), n, n_samples, H.ptr(rgb), H.ptr(dmap), H.ptr(acc),
                                                 H.stream_ptr(dev)), "nf_render_volume_density")
        ctx.model, ctx.n_samples = model, n_samples
        ctx.save_for_backward(raw, depth, saved)
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        raw, depth, saved = ctx.saved_tensors
        model, n_samples = ctx.model, ctx.n_samples
        dev = raw.device
        n = raw.shape[0]
        lib = H.lib()
        d_rgb = _c(d_rgb)
        d_raw = torch.empty_like(raw)
        with torch.cuda.device(dev):
            ws_n = _fn(model, "bwd_workspace_floats")(n * n_samples)
        ws = torch.empty(ws_n, dtype=torch.float32, device=dev)
        flat = torch.empty(_fn(model, "grad_floats")(), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            H.check(lib.nf_render_volume_density_bwd(H.ptr(raw), H.ptr(depth), H.ptr(d_rgb), n, n_samples, H.ptr(d_raw), H.stream_ptr(dev)),
                    "nf_render_volume_density_bwd")
            H.check(_fn(model, "mlp_bwd")(H.ptr(_image(model, "pack_bwd")), H.ptr(saved), H.ptr(d_raw), n, n_samples, H.ptr(ws), ws_n,
                                          H.ptr(flat), H.stream_ptr(dev)), "mlp_bwd")
        grads, off = [], 0
        for p in model.hip_param_list():
            grads.append(flat[off:off + p.numel()].view(p.shape))
            off += p.numel()
        return (None, None, None, None, None, *grads)


def _render_image(height, width, ray_origins, ray_directions, depth, depth_samples_per_ray, model):
    """Shared tail of run_one_iter_of_tinynerf: ray bundle + per-ray depths (H, W, S) -> fused MLP -> compositing."""
    dev = ray_origins.device
    ro, rd = ray_origins.reshape(-1, 3), ray_directions.reshape(-1, 3)
    depth2 = _c(depth.reshape(-1, depth_samples_per_ray))
    n = ro.shape[0]
    if torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()):
        rgb = _TinyRender.apply(model, _c(ro), _c(rd), depth2, int(depth_samples_per_ray), *model.hip_param_list())
        return rgb.reshape(height, width, 3)
    raw = torch.empty((n, depth_samples_per_ray, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        H.check(_fn(model, "mlp_fwd")(H.ptr(_image(model, "pack")), H.ptr(_c(ro)), H.ptr(_c(rd)), H.ptr(depth2), 1, n,
                                      depth_samples_per_ray, H.ptr(raw), H.stream_ptr(dev)), "mlp_fwd")
    rgb, _, _ = render_volume_density(raw.reshape(height, width, depth_samples_per_ray, 4), ray_origins, depth)
    return rgb


class GraphedTinyTrainer:
    """One training iteration of the tiny path (the loop body of TN:282-302: rgb = run_one_iter_of_tinynerf(...), loss =
    mse(rgb, target), loss.backward(), optimizer.step(), optimizer.zero_grad()) captured ONCE in a HIP graph and replayed.

    An iteration is ~25 launches of 5-100 us each (ray bundle, jitter, fused MLP forward with saves, compositing, loss, the
    backward kernels, Adam), i.e. bound by the host's launch rate, not by the device: 2.1 ms eager against the ~0.3 ms the
    kernels take.  The graph removes the host from the loop.  Not in the reference (an MI355X extension); differences to the
    eager call: the depth jitter is drawn ON THE DEVICE inside the graph (the reference draws torch.rand on the host and copies
    it, TN:46-57 -- a host-to-device copy cannot be captured), or supplied by the caller through `jitter`; pose and target are
    copied into static buffers before each replay.  The optimizer must be capturable (torch.optim.Adam(..., capturable=True)).
    """

    def __init__(self, model, optimizer, height, width, focal_length, near_thresh, far_thresh, depth_samples_per_ray, device,
                 warmup: int = 3):
        _api(model)                                                                  # raises for a model without kernels
        self.model, self.optimizer = model, optimizer
        self.h, self.w, self.focal, self.s = int(height), int(width), focal_length, int(depth_samples_per_ray)
        self.near, self.far = float(near_thresh), float(far_thresh)
        dev = torch.device(device)
        self.pose = torch.zeros((4, 4), dtype=torch.float32, device=dev)
        self.target = torch.zeros((self.h, self.w, 3), dtype=torch.float32, device=dev)
        self.jitter = torch.zeros((self.h, self.w, self.s), dtype=torch.float32, device=dev)
        self.t_vals = torch.linspace(self.near, self.far, self.s, device=dev)        # TN:46 (torch.linspace rounds the same on both devices)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self._own_jitter = True
        self.graph = None
        self._warmup = int(warmup)

    def _iteration(self):
        if self._own_jitter:
            self.jitter.uniform_(0.0, 1.0)                                           # TN:52-57, on the device
        depth = self.t_vals + self.jitter * (self.far - self.near) / self.s
        bump_pack_epoch()
        ray_origins, ray_directions = get_ray_bundle(self.h, self.w, self.focal, self.pose)
        rgb = _render_image(self.h, self.w, ray_origins, ray_directions, depth, self.s, self.model)
        loss = torch.nn.functional.mse_loss(rgb, self.target)
        loss.backward()
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=False)
        self.loss.copy_(loss.detach())

    def _capture(self):
        dev = self.pose.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                                                # warm-up on a side stream, as graph capture requires
            for _ in range(self._warmup):
                self._iteration()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._iteration()

    def step(self, tform_cam2world, target_img, jitter: Optional[torch.Tensor] = None):
        """One iteration on (pose, target image); returns the loss (a device scalar that the NEXT call overwrites).
        jitter (H, W, S) in [0, 1): the caller's random numbers instead of the device draw (fixed at the first call)."""
        own = jitter is None
        if self.graph is None:
            self._own_jitter = own
            self.pose.copy_(tform_cam2world, non_blocking=True)
            self.target.copy_(target_img, non_blocking=True)
            if not own:
                self.jitter.copy_(jitter, non_blocking=True)
            self._capture()                                                          # note: the warm-up iterations train, too
        if own != self._own_jitter:
            raise ValueError("GraphedTinyTrainer: the jitter source is fixed when the graph is captured (first call)")
        self.pose.copy_(tform_cam2world, non_blocking=True)
        self.target.copy_(target_img, non_blocking=True)
        if not own:
            self.jitter.copy_(jitter, non_blocking=True)
        self.graph.replay()
        return self.loss


def run_one_iter_of_tinynerf(height, width, focal_length, tform_cam2world, near_thresh, far_thresh, depth_samples_per_ray,
                             encoding_function, get_minibatches_function, chunksize, model, encoding_function_args):
    """TN:111-159 (same signature; `encoding_function`, `get_minibatches_function` and `chunksize` are accepted for
    compatibility -- encoding and chunking happen inside the fused kernel).  Returns rgb_predicted (H, W, 3)."""
    _api(model)                        # raises for a model without kernels
    if int(encoding_function_args) != 10:
        raise NotImplementedError("the fused tiny kernels encode with num_encoding_functions=10 (TN:276)")
    bump_pack_epoch()                  # weight images are rebuilt once per call (fused optimizers do not bump version counters)
    ray_origins, ray_directions = get_ray_bundle(height, width, focal_length, tform_cam2world)
    depth_values = _depths(ray_origins, near_thresh, far_thresh, depth_samples_per_ray, True)       # default randomize=True
    return _render_image(height, width, ray_origins, ray_directions, depth_values, depth_samples_per_ray, model)
