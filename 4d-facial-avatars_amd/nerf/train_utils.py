"""MI355X-native render orchestration: the reference's nerf/train_utils.py behind the same API.

`run_one_iter_of_nerf` keeps the reference signature and return tuple (T:165-290) but, instead of the
reference's chain of stock tensor ops that materialises every intermediate (T:9-33, T:36-162), it drives
the HIP kernels of libnerface_hip.so per ray chunk:

    K2 coarse depths -> K4 fused MLP (coarse) -> K5 integrator -> K6+K7 resample/merge
                     -> K4 fused MLP (fine)   -> K5 integrator

Ray chunking (`chunksize` rays per chunk, T:229) and the order/shape of the random draws per chunk
(t_rand, coarse noise, u, fine noise -- SURVEY §8 A4) are kept, so a run with the same torch seed on the
same device consumes the RNG stream exactly as the reference would.  Point chunking (T:20) has no
counterpart: the fused kernel tiles points internally and its results do not depend on a chunk size.

Citations: T = nerf/train_utils.py, V = nerf/volume_rendering_utils.py, M = nerf/models.py (reference).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .nerf_helpers import PositionalEncoder, get_minibatches
from .volume_rendering_utils import volume_render_radiance_field  # noqa: F401  (re-exported like the reference)


def run_network(network_fn, pts, ray_batch, chunksize, embed_fn, embeddirs_fn, expressions=None, latent_code=None):
    """T:9-33, the unfused form: encode points and "view directions" (= ray_batch[..., -3:], Quirk Q1), concatenate, evaluate
    the model in point chunks, reshape.  Kept for API compatibility (inference; kernels K3 + nf_paper_forward_encoded);
    run_one_iter_of_nerf does all of this inside one fused kernel without materialising the encodings."""
    pts_flat = pts.reshape((-1, pts.shape[-1]))
    embedded = embed_fn(pts_flat)
    if embeddirs_fn is not None:
        viewdirs = ray_batch[..., None, -3:]
        input_dirs_flat = viewdirs.expand(pts.shape).reshape((-1, 3))
        embedded = torch.cat((embedded, embeddirs_fn(input_dirs_flat.contiguous())), dim=-1)
    batches = get_minibatches(embedded, chunksize=chunksize)
    if expressions is None:
        preds = [network_fn(batch) for batch in batches]
    elif latent_code is not None:
        preds = [network_fn(batch, expressions, latent_code) for batch in batches]
    else:
        preds = [network_fn(batch, expressions) for batch in batches]
    radiance_field = torch.cat(preds, dim=0)
    return radiance_field.reshape(list(pts.shape[:-1]) + [radiance_field.shape[-1]])


# --------------------------------------------------------------------------------------------------
# autograd boundary: one Function per ray chunk
# --------------------------------------------------------------------------------------------------
class _RenderChunk(torch.autograd.Function):
    """Coarse+fine render of one ray chunk.  Inputs that can receive gradients: the 2x26 model parameters
    and the latent code (nothing upstream of the MLP is learnable).  Outputs: the 7-tuple of
    T:162; only rgb_coarse / rgb_fine are differentiable (the trainer's loss uses nothing else)."""

    @staticmethod
    def forward(ctx, cfg, ro, rd, rd_view, bg, expr, latent, t_rand, noise_c, u, noise_f, n_params_c, *params):
        model_c, model_f = cfg["model_coarse"], cfg["model_fine"]
        near, far, nc, nf = cfg["near"], cfg["far"], cfg["num_coarse"], cfg["num_fine"]
        white = cfg["white_background"]
        need_grad = cfg["need_grad"]
        ctx.set_materialize_grads(False)       # the five non-differentiable outputs would each cost a zero-fill kernel per backward
        dev = ro.device
        n_rays = ro.shape[0]

        lat_d = latent.detach().reshape(-1).contiguous()
        z_c = ops.sample_coarse(n_rays, nc, near, far, dev, t_rand, lindisp=cfg["lindisp"])
        raw_c, state_c = model_c.hip_forward(ro, rd, z_c, rd_view, expr, lat_d, near, far, need_grad)
        rgb_c, disp_c, acc_c, w_c = ops.volume_render_fwd(raw_c, z_c, rd, noise_c, bg, white)
        outs = [rgb_c, disp_c, acc_c]
        ctx.has_fine = nf > 0 and model_f is not None
        if ctx.has_fine:
            z_f = ops.resample_merge(z_c, w_c, nf, u)
            raw_f, state_f = model_f.hip_forward(ro, rd, z_f, rd_view, expr, lat_d, near, far, need_grad)
            rgb_f, disp_f, acc_f, w_f = ops.volume_render_fwd(raw_f, z_f, rd, noise_f, bg, white)
            w_last = w_f[:, -1].contiguous()
            outs += [rgb_f, disp_f, acc_f, w_last]
        else:
            w_last = w_c[:, -1].contiguous()
            outs += [w_last]
        if need_grad:
            ctx.cfg = cfg
            ctx.n_params_c = n_params_c
            ctx.save_for_backward(rd, bg, noise_c, noise_f, z_c, raw_c, latent)
            ctx.state_c = state_c
            if ctx.has_fine:
                ctx.fine = (z_f, raw_f, state_f)
        nondiff = [o for i, o in enumerate(outs) if not (i == 0 or (ctx.has_fine and i == 3))]
        ctx.mark_non_differentiable(*nondiff)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        cfg = ctx.cfg
        rd, bg, noise_c, noise_f, z_c, raw_c, latent = ctx.saved_tensors
        model_c, model_f = cfg["model_coarse"], cfg["model_fine"]
        white = cfg["white_background"]
        d_rgb_c = grads[0]
        g_latent = None
        grads_c = [None] * ctx.n_params_c
        grads_f = []
        if d_rgb_c is not None:
            d_raw_c = ops.volume_render_bwd(raw_c, z_c, rd, noise_c, bg, d_rgb_c, white)
            grads_c, g_latent = model_c.hip_backward(ctx.state_c, z_c, d_raw_c)
        ctx.state_c = None
        if ctx.has_fine:
            z_f, raw_f, state_f = ctx.fine
            d_rgb_f = grads[3]
            grads_f = [None] * len(model_f.hip_param_list())
            if d_rgb_f is not None:
                d_raw_f = ops.volume_render_bwd(raw_f, z_f, rd, noise_f, bg, d_rgb_f, white)
                grads_f, gl = model_f.hip_backward(state_f, z_f, d_raw_f)
                g_latent = gl if g_latent is None else g_latent + gl
            ctx.fine = None
        g_latent = g_latent.reshape(latent.shape) if (ctx.needs_input_grad[6] and g_latent is not None) else None
        # inputs: cfg, ro, rd, rd_view, bg, expr, latent, t_rand, noise_c, u, noise_f, n_params_c, *params
        return (None, None, None, None, None, None, g_latent, None, None, None, None, None, *grads_c, *grads_f)


def _check_encoders(encode_position_fn, encode_direction_fn):
    ok = (isinstance(encode_position_fn, PositionalEncoder) and encode_position_fn.num_encoding_functions == 10
          and encode_position_fn.include_input and encode_position_fn.log_sampling
          and isinstance(encode_direction_fn, PositionalEncoder) and encode_direction_fn.num_encoding_functions == 4
          and not encode_direction_fn.include_input and encode_direction_fn.log_sampling)
    if not ok:
        raise NotImplementedError(
            "the fused MLP kernel is built for the NeRFace encoders: get_embedding_function(10, include_input=True) "
            "for positions and get_embedding_function(4, include_input=False) for directions (log sampling)")


def _check_chunk_args(model_coarse, encode_position_fn, encode_direction_fn, expressions, latent_code, rays):
    _check_encoders(encode_position_fn, encode_direction_fn)
    if expressions is None or latent_code is None:
        raise NotImplementedError("the NeRFace path needs `expressions` and `latent_code` (unconditioned NeRF is out of scope)")
    if not getattr(model_coarse, "fused_supported", lambda: False)():
        raise NotImplementedError(f"{type(model_coarse).__name__}: no fused HIP kernel for this model/geometry")
    if not rays.is_cuda:
        raise RuntimeError("nerf (MI355X build): rays must be on a ROCm device; there is no CPU path")


def predict_and_render_radiance(ray_batch, model_coarse, model_fine, options, mode="train", encode_position_fn=None,
                                encode_direction_fn=None, expressions=None, background_prior=None, latent_code=None,
                                ray_dirs_fake=None):
    """T:36-162 for one ray chunk: ray_batch (n, 8) = [ro, rd, near, far]; returns the 7-tuple of T:162."""
    _check_chunk_args(model_coarse, encode_position_fn, encode_direction_fn, expressions, latent_code, ray_batch)
    n_rays = ray_batch.shape[0]
    rb = ray_batch.to(torch.float32)
    ro = rb[:, 0:3].contiguous()
    rd = rb[:, 3:6].contiguous()
    # Quirk Q7 (T:81-82): on the ablation path the *encoded* direction comes from chunk 0 of the fake rays
    rd_view = None
    if ray_dirs_fake:
        rd_view = ray_dirs_fake[0][:n_rays, 3:6].to(torch.float32).contiguous()
        ray_batch[..., 3:6] = ray_dirs_fake[0][..., 3:6]           # same in-place side effect as the reference
    return _render_rays(ro, rd, rd_view, model_coarse, model_fine, options, mode, expressions, background_prior, latent_code)


def _render_rays(ro, rd, rd_view, model_coarse, model_fine, options, mode, expressions, background_prior, latent_code):
    """The body of predict_and_render_radiance on (n, 3) origins / directions: run_one_iter_of_nerf calls it with row views of
    its inputs, without the (n, 8) ray_batch detour of T:206-212 (two fills, two multiplies, a cat and two slice copies per call)."""
    m = getattr(options.nerf, mode)
    dev = ro.device
    n_rays = ro.shape[0]
    nc, nf = int(m.num_coarse), int(m.num_fine)
    near, far = float(options.dataset.near), float(options.dataset.far)
    has_fine = nf > 0 and model_fine is not None
    noise_std = float(m.radiance_field_noise_std)
    # ---- random draws, in the reference's order and shapes (T:75, V:41-50, H:363-367) --------------
    t_rand = torch.rand((n_rays, nc), dtype=torch.float32, device=dev) if m.perturb else None
    noise_c = (torch.randn((n_rays, nc), dtype=torch.float32, device=dev) * noise_std) if noise_std > 0.0 else None
    u = noise_f = None
    if nf > 0:
        det = (m.perturb == 0.0)
        u = None if det else torch.rand([n_rays, nf], dtype=torch.float32, device=dev)
        if model_fine is not None and noise_std > 0.0:
            noise_f = torch.randn((n_rays, nc + nf), dtype=torch.float32, device=dev) * noise_std
    bg = None
    if background_prior is not None:
        bg = background_prior.to(device=dev, dtype=torch.float32).reshape(-1, 3).contiguous()
    params_c = model_coarse.hip_param_list()
    params_f = model_fine.hip_param_list() if has_fine else []
    expr = expressions.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
    need_grad = torch.is_grad_enabled() and (latent_code.requires_grad or any(p.requires_grad for p in params_c + params_f))
    cfg = dict(model_coarse=model_coarse, model_fine=model_fine if has_fine else None, near=near, far=far, num_coarse=nc,
               lindisp=bool(m.lindisp),
               num_fine=nf if has_fine else 0, white_background=bool(m.white_background), need_grad=need_grad)
    lat = latent_code if latent_code.dtype == torch.float32 else latent_code.to(torch.float32)
    outs = _RenderChunk.apply(cfg, ro, rd, rd_view, bg, expr, lat, t_rand, noise_c, u, noise_f, len(params_c),
                              *params_c, *params_f)
    if has_fine:
        return tuple(outs)
    rgb_c, disp_c, acc_c, w_last = outs
    return rgb_c, disp_c, acc_c, None, None, None, w_last


def run_one_iter_of_nerf(height, width, focal_length, model_coarse, model_fine, ray_origins, ray_directions, options,
                         mode="train", encode_position_fn=None, encode_direction_fn=None, expressions=None,
                         background_prior=None, latent_code=None, ray_directions_ablation=None):
    """T:165-290.  Same signature; returns (rgb_coarse, disp_coarse, acc_coarse, rgb_fine, disp_fine, acc_fine,
    weights_fine[:, -1]), reshaped to image planes in `validation` mode (T:275-284)."""
    is_rad = torch.is_tensor(ray_directions_ablation)
    ops.bump_pack_epoch()              # weight images are rebuilt once per call: see ops.PaperWeights (fused optimizers, .data writes)
    if options.dataset.no_ndc is False:
        raise NotImplementedError("NDC rays are not part of the NeRFace path (all configs use no_ndc: True)")
    restore_shapes = [ray_directions.shape, ray_directions.shape[:-1], ray_directions.shape[:-1]]
    if model_fine:
        restore_shapes += restore_shapes
        restore_shapes += [ray_directions.shape[:-1]]
    ro = ray_origins.reshape((-1, 3))
    rd = ray_directions.reshape((-1, 3))
    chunksize = getattr(options.nerf, mode).chunksize
    bg_chunks = get_minibatches(background_prior, chunksize=chunksize) if background_prior is not None else None
    if is_rad:
        # ablation call pattern (EV:449-467): the reference's own (R, 8) ray batches, chunked, with the in-place side effect of
        # T:81-82 on them
        near = options.dataset.near * torch.ones_like(rd[..., :1])
        far = options.dataset.far * torch.ones_like(rd[..., :1])
        rays = torch.cat((ro, rd, near, far), dim=-1)                              # (R, 8), T:206-212
        batches = get_minibatches(rays, chunksize=chunksize)
        rays_ablation = torch.cat((ro, ray_directions_ablation.reshape((-1, 3)), near, far), dim=-1)
        batches_ablation = get_minibatches(rays_ablation, chunksize=chunksize)
        pred = [
            predict_and_render_radiance(batch, model_coarse, model_fine, options, mode, encode_position_fn=encode_position_fn,
                                        encode_direction_fn=encode_direction_fn, expressions=expressions,
                                        background_prior=bg_chunks[i] if bg_chunks is not None else None,
                                        latent_code=latent_code, ray_dirs_fake=batches_ablation)
            for i, batch in enumerate(batches)
        ]
    else:
        # same chunks (T:213 get_minibatches over the rays), taken as row views of the caller's origins / directions: near and far
        # are the two scalars of the config, so the (R, 8) concatenation of T:206-212 carries nothing the kernels need
        _check_chunk_args(model_coarse, encode_position_fn, encode_direction_fn, expressions, latent_code, ro)
        ro32 = ro if (ro.dtype == torch.float32 and ro.is_contiguous()) else ro.to(torch.float32).contiguous()
        rd32 = rd if (rd.dtype == torch.float32 and rd.is_contiguous()) else rd.to(torch.float32).contiguous()
        pred = [
            _render_rays(ro32[i0:i0 + chunksize], rd32[i0:i0 + chunksize], None, model_coarse, model_fine, options, mode, expressions,
                         bg_chunks[k] if bg_chunks is not None else None, latent_code)
            for k, i0 in enumerate(range(0, ro32.shape[0], chunksize))
        ]
    synthesized = list(zip(*pred))
    synthesized = [(img[0] if len(img) == 1 else torch.cat(img, dim=0)) if img[0] is not None else None for img in synthesized]
    if ops.get_mlp_precision() in ops.F16_MODES and not torch.is_grad_enabled():
        ops.check_f16_range(model_coarse, model_fine)              # once per call (per frame in validation mode), after all chunks
    if mode == "validation":
        synthesized = [img.view(shape) if img is not None else None for (img, shape) in zip(synthesized, restore_shapes)]
        if model_fine:
            return tuple(synthesized)
        return tuple(synthesized + [None, None, None])
    return tuple(synthesized)


class GaussianSmoothing(torch.nn.Module):
    """T:379-442: depthwise Gaussian blur used only by commented-out experiments of the trainer.  Kept so that
    `from nerf import GaussianSmoothing` succeeds (train_transformed_rays.py:19-21), with the reference's own kernel
    formula exp(-((x - mean) / (2 sigma))^2) (T:409-410 -- not the textbook exp(-x^2 / (2 sigma^2)); a drop-in keeps it)
    and its fixed padding of 5 (T:442)."""

    def __init__(self, channels, kernel_size, sigma, dim=2):
        super().__init__()
        import math
        import numbers
        if isinstance(kernel_size, numbers.Number):
            kernel_size = [kernel_size] * dim
        if isinstance(sigma, numbers.Number):
            sigma = [sigma] * dim
        kernel = 1
        grids = torch.meshgrid([torch.arange(s, dtype=torch.float32) for s in kernel_size], indexing="ij")
        for size, std, mgrid in zip(kernel_size, sigma, grids):
            mean = (size - 1) / 2
            kernel = kernel * (1 / (std * math.sqrt(2 * math.pi)) * torch.exp(-((mgrid - mean) / (2 * std)) ** 2))
        kernel = kernel / torch.sum(kernel)
        kernel = kernel.view(1, 1, *kernel.size())
        kernel = kernel.repeat(channels, *[1] * (kernel.dim() - 1))
        self.register_buffer("weight", kernel)
        self.groups = channels
        self.conv = {1: torch.nn.functional.conv1d, 2: torch.nn.functional.conv2d, 3: torch.nn.functional.conv3d}[dim]

    def forward(self, input):
        return self.conv(input, weight=self.weight, groups=self.groups, padding=5)
