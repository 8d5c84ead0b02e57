"""Model classes of the reference's nerf/models.py that lie on the NeRFace hot path.

`ConditionalBlendshapePaperNeRFModel` (reference nerf/models.py:189-261, the model of 88 of the 108
`type:` entries under config/) keeps the reference's constructor signature, parameter names, shapes and
default initialisation, so checkpoints (`model_{coarse,fine}_state_dict`) load unchanged.  Its compute is
the fused HIP kernel K4: `run_one_iter_of_nerf` hands the module to the kernel wrapper, which reads the
live parameter storages (no copies besides the cached fragment-ordered image).
"""
from __future__ import annotations

import torch

from . import ops


class ConditionalBlendshapePaperNeRFModel(torch.nn.Module):
    r"""NeRFace paper model (Fig. 7 of the paper; reference nerf/models.py:189-261).

    x0 = [PE10(xyz) (63) | expression*1/3 (76) | latent code (32)] -> 3 x (Linear 256 + ReLU) ->
    [x0 | h] -> 3 x (Linear 256 + ReLU) -> feat = fc_feat(h) -> sigma = fc_alpha(feat);
    [feat | PE4(dir) (24)] -> 3 x (Linear 128 + ReLU) -> rgb = fc_rgb.  `layers_dir.3` exists in every
    checkpoint but is never used (Quirk Q3); num_layers / hidden_size / skip_connect_every are accepted
    and ignored exactly as in the reference (widths are hard-coded).
    """

    def __init__(self, num_layers=8, hidden_size=256, skip_connect_every=4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4,
                 include_input_xyz=True, include_input_dir=True, use_viewdirs=True, include_expression=True,
                 latent_code_dim=32):
        super().__init__()
        include_input_xyz = 3 if include_input_xyz else 0
        include_input_dir = 3 if include_input_dir else 0
        include_expression = 76 if include_expression else 0
        self.dim_xyz = include_input_xyz + 2 * 3 * num_encoding_fn_xyz
        self.dim_dir = include_input_dir + 2 * 3 * num_encoding_fn_dir
        self.dim_expression = include_expression
        self.dim_latent_code = latent_code_dim
        self.use_viewdirs = use_viewdirs
        d_in = self.dim_xyz + self.dim_expression + self.dim_latent_code
        self.layers_xyz = torch.nn.ModuleList()
        self.layers_xyz.append(torch.nn.Linear(d_in, 256))
        for i in range(1, 6):
            self.layers_xyz.append(torch.nn.Linear(d_in + 256 if i == 3 else 256, 256))
        self.fc_feat = torch.nn.Linear(256, 256)
        self.fc_alpha = torch.nn.Linear(256, 1)
        self.layers_dir = torch.nn.ModuleList()
        self.layers_dir.append(torch.nn.Linear(256 + self.dim_dir, 128))
        for _ in range(3):
            self.layers_dir.append(torch.nn.Linear(128, 128))
        self.fc_rgb = torch.nn.Linear(128, 3)
        self.relu = torch.nn.functional.relu
        self._hip_weights = None

    # ---- kernel plumbing -------------------------------------------------------------------------
    def fused_supported(self) -> bool:
        """The HIP kernel is specialised to the one geometry every NeRFace config instantiates."""
        return (self.dim_xyz == 63 and self.dim_dir == 24 and self.dim_expression == 76 and self.dim_latent_code == 32
                and self.use_viewdirs)

    def hip_param_list(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in ops.PAPER_KEYS]

    def hip_weights(self) -> "ops.PaperWeights":
        params = self.hip_param_list()
        hw = self._hip_weights
        if hw is None or any(a is not b for a, b in zip(hw._params, params)):
            hw = ops.PaperWeights(params)
            self._hip_weights = hw
        return hw

    # ---- what run_one_iter_of_nerf calls (one interface for every fused model family) -------------------------
    def hip_forward(self, ro, rd, z, rd_view, expr, latent, near, far, need_grad):
        """raw (R, S, 4) for the points ro + rd*z; `state` is what hip_backward needs (None when no gradient is wanted)."""
        hw = self.hip_weights()
        packed = hw.get()
        cond = ops.paper_condition(packed, expr, latent, near, far)
        if need_grad:
            ops.require_trainable_precision()
            prec = ops.get_mlp_precision()
            pb = hw.get_bf16() if prec == "bf16x3" else None
            ph = hw.get_f16() if prec == "f16x3" else None
            if ph is not None:
                # range probe (see the inference branch): weights move every step, so probe the first call and every 128th
                n_calls = self.__dict__["_f16_train_calls"] = self.__dict__.get("_f16_train_calls", 0) + 1
                if ops.f16_train_probe_every() == 1 or n_calls % ops.f16_train_probe_every() == 1:
                    amax = ops.f16_preflight(self, ro, rd, z, rd_view, expr, latent, near, far)
                    if not amax * ops.F16_PREFLIGHT_MARGIN < ops.F16_ACT_LIMIT:
                        raise RuntimeError(f'nerf.set_mlp_precision("f16x3"): hidden activations of {type(self).__name__} reach {amax:.3g}, '
                                           f'within {ops.F16_PREFLIGHT_MARGIN:g}x of the fp16 range limit ({ops.F16_ACT_LIMIT:g}) -- train '
                                           f'this model with "f32" or "bf16x3"')
            raw, saved = ops.paper_mlp_fwd_train(packed, cond, ro, rd, z, rd_view, packed_b=pb, packed_h=ph)
            return raw, (packed, cond, saved, "f16" if ph is not None else pb is not None)
        if ops.get_mlp_precision() == "bf16x3":
            return ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, ro, rd, z, rd_view), None
        if ops.get_mlp_precision() in ops.F16_MODES:
            key = (hw._signature()[1:], expr.data_ptr(), latent.data_ptr(), expr._version, latent._version)
            if getattr(self, "_f16_probe_key", None) != key:      # once per (weights, conditioning): i.e. once per frame and model
                amax = ops.f16_preflight(self, ro, rd, z, rd_view, expr, latent, near, far)
                if not amax * ops.F16_PREFLIGHT_MARGIN < ops.F16_ACT_LIMIT:
                    raise RuntimeError(f'nerf.set_mlp_precision("{ops.get_mlp_precision()}"): hidden activations of {type(self).__name__} reach {amax:.3g} on a '
                                       f'sample of this frame, within {ops.F16_PREFLIGHT_MARGIN:g}x of the fp16 range limit '
                                       f'({ops.F16_ACT_LIMIT:g}) -- render this model with "f32" or "bf16x3"')
                self._f16_probe_key = key
            if ops.get_mlp_precision() == "f16x2":
                return ops.paper_mlp_fwd_f16x2(hw.get_f16(), cond, ro, rd, z, rd_view), None
            return ops.paper_mlp_fwd_f16(hw.get_f16(), cond, ro, rd, z, rd_view), None
        return ops.paper_mlp_fwd(packed, cond, ro, rd, z, rd_view), None

    def hip_backward(self, state, z, d_raw):
        """d_raw -> ([gradients in hip_param_list() order, None for layers_dir.3], d_latent (32))."""
        packed, cond, saved, split = state
        return ops.paper_mlp_bwd(self, packed, cond, z, d_raw, saved, split=split)

    def forward(self, x, expr=None, latent_code=None, **kwargs):
        """M:236-261 on pre-encoded inputs x (N, 87) = [PE10(xyz) | PE4(dirs)] -> (N, 4), as run_network calls it (T:20-24).
        Inference only (kernel nf_paper_forward_encoded); training goes through run_one_iter_of_nerf, whose fused kernels
        own the backward."""
        from . import _hip as H
        if not self.fused_supported() or expr is None or latent_code is None:
            raise NotImplementedError("forward() is built for the NeRFace geometry and needs expr and latent_code")
        if torch.is_grad_enabled() and (x.requires_grad or latent_code.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("ConditionalBlendshapePaperNeRFModel.forward has no autograd on the MI355X build: train through "
                                      "nerf.run_one_iter_of_nerf(...), or call forward under torch.no_grad()")
        x = ops._c(x.detach())
        if x.dim() != 2 or x.shape[1] != 87:
            raise ValueError("expected pre-encoded inputs of shape (N, 87)")
        expr_d, lat_d = ops._c(expr.detach()).reshape(-1), ops._c(latent_code.detach()).reshape(-1)
        packed = self.hip_weights().get()
        dev = H.require_device(packed, x, expr_d, lat_d)
        lib = H.lib()
        cond = torch.empty(lib.nf_paper_cond_floats(), dtype=torch.float32, device=dev)
        out = torch.empty((x.shape[0], 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            H.check(lib.nf_paper_forward_encoded(H.ptr(packed), H.ptr(x), H.ptr(expr_d), H.ptr(lat_d), x.shape[0], H.ptr(cond),
                                                 H.ptr(out), H.stream_ptr(dev)), "nf_paper_forward_encoded")
        return out


LCODE_KEYS = [f"{n}.{p}" for n in ("layer1", "layers_xyz.0", "layers_xyz.1", "layers_xyz.2", "layers_dir.0", "fc_alpha", "fc_rgb", "fc_feat")
              for p in ("weight", "bias")]


class ConditionalBlendshapeLearnableCodeNeRFModel(torch.nn.Module):
    r"""Second NeRFace model family (reference nerf/models.py:529-636; 6 config entries): layer1 without activation, three
    256-wide ReLU layers, feat = relu(fc_feat(x)), sigma = fc_alpha(x), one 280 -> 128 direction layer, fc_rgb.
    Same constructor signature, parameter names and shapes as the reference, so its checkpoints load.  The MI355X build
    provides the forward and the backward (exact-f32 fused kernels) for the geometry the configs use: num_layers=4,
    hidden_size=256 (the trainer never passes skip_connect_every, TR:100-109), 10/4 encoding functions."""

    def __init__(self, num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4,
                 include_input_xyz=True, include_input_dir=True, use_viewdirs=True, include_expression=True, latent_code_dim=32):
        super().__init__()
        include_input_xyz = 3 if include_input_xyz else 0
        include_input_dir = 3 if include_input_dir else 0
        include_expression = 76 if include_expression else 0
        self.dim_xyz = include_input_xyz + 2 * 3 * num_encoding_fn_xyz
        self.dim_dir = include_input_dir + 2 * 3 * num_encoding_fn_dir if use_viewdirs else 0
        self.dim_expression = include_expression
        self.skip_connect_every = skip_connect_every
        self.dim_latent_code = latent_code_dim
        self.layers_expr = None
        d_in = self.dim_xyz + self.dim_expression + self.dim_latent_code
        self.layer1 = torch.nn.Linear(d_in, hidden_size)
        self.layers_xyz = torch.nn.ModuleList()
        for i in range(num_layers - 1):
            skip = i % self.skip_connect_every == 0 and i > 0 and i != num_layers - 1
            self.layers_xyz.append(torch.nn.Linear(self.dim_xyz + hidden_size + self.dim_expression + self.dim_latent_code if skip
                                                   else hidden_size, hidden_size))
        self.use_viewdirs = use_viewdirs
        if self.use_viewdirs:
            self.layers_dir = torch.nn.ModuleList()
            self.layers_dir.append(torch.nn.Linear(self.dim_dir + hidden_size, hidden_size // 2))
            self.fc_alpha = torch.nn.Linear(hidden_size, 1)
            self.fc_rgb = torch.nn.Linear(hidden_size // 2, 3)
            self.fc_feat = torch.nn.Linear(hidden_size, hidden_size)
        else:
            self.fc_out = torch.nn.Linear(hidden_size, 4)
        self.relu = torch.nn.functional.relu
        self.sigmoid = torch.sigmoid

    def fused_supported(self) -> bool:
        return (self.use_viewdirs and self.dim_xyz == 63 and self.dim_dir == 24 and self.dim_expression == 76
                and self.dim_latent_code == 32 and self.layer1.out_features == 256 and len(self.layers_xyz) == 3
                and all(l.in_features == 256 for l in self.layers_xyz))

    def hip_param_list(self):
        sd = dict(self.named_parameters())
        return [sd[k] for k in LCODE_KEYS]

    # one cached image per kind: (size function, pack function, element dtype); rebuilt when a parameter's version moves
    _PACK_KINDS = {
        "f32": ("nf_lcode_packed_floats", "nf_lcode_pack", torch.float32),
        "f32_t": ("nf_lcode_packed_bwd_floats", "nf_lcode_pack_bwd", torch.float32),
        "bf16": ("nf_lcode_packed_bf16_bytes", "nf_lcode_pack_bf16", torch.uint8),
        "bf16_t": ("nf_lcode_packed_bwd_bf16_bytes", "nf_lcode_pack_bwd_bf16", torch.uint8),
        "f16": ("nf_lcode_packed_f16_bytes", "nf_lcode_pack_f16", torch.uint8),
        "f16_t": ("nf_lcode_packed_bwd_f16_bytes", "nf_lcode_pack_bwd_f16", torch.uint8),
    }

    def _hip_pack(self, kind):
        import ctypes as C
        from . import _hip as H
        ps = self.hip_param_list()
        sig = (ops.pack_epoch(),) + tuple((int(p.data_ptr()), int(p._version)) for p in ps)
        cache = self.__dict__.setdefault("_pack_cache", {})
        hit = cache.get(kind)
        if hit is None or hit[0] != sig:
            size_fn, pack_fn, dtype = self._PACK_KINDS[kind]
            dev = H.require_device(*[p.detach() for p in ps])
            lib = H.lib()
            buf = hit[1] if hit is not None and hit[1].device == dev else torch.empty(getattr(lib, size_fn)(), dtype=dtype, device=dev)
            if kind == "f16" and hit is not None and hit[1] is buf:
                # packing clears the stream's range-guard flag: carry it over first (cf. ops.PaperWeights._get)
                sticky = self.__dict__.get("_f16_sticky")
                if sticky is None or sticky.device != dev:
                    sticky = self.__dict__["_f16_sticky"] = torch.zeros((), dtype=torch.int32, device=dev)
                off = lib.nf_lcode_f16_flag_offset()
                sticky.bitwise_or_(buf[off:off + 4].view(torch.int32)[0])
            arr = (C.c_void_p * len(ps))(*[int(p.data_ptr()) for p in ps])
            with torch.cuda.device(dev):
                H.check(getattr(lib, pack_fn)(arr, H.ptr(buf), H.stream_ptr(dev)), pack_fn)
            cache[kind] = (sig, buf)
        return cache[kind][1]

    def _hip_packed(self):
        return self._hip_pack("f32")

    def _hip_packed_t(self):
        return self._hip_pack("f32_t")

    def _hip_packed_bf16(self):
        return self._hip_pack("bf16")

    def _hip_packed_bf16_t(self):
        return self._hip_pack("bf16_t")

    def hip_weights(self):
        """The interface ops.check_f16_range polls (range-guard flag of the split-fp16 stream)."""
        model = self

        class _W:
            def f16_range_flag(self):
                from . import _hip as H
                hit = model.__dict__.get("_pack_cache", {}).get("f16")
                if hit is None:
                    return None
                off = H.lib().nf_lcode_f16_flag_offset()
                flag = hit[1][off:off + 4].view(torch.int32)[0]
                sticky = model.__dict__.get("_f16_sticky")
                return flag if sticky is None else torch.bitwise_or(flag, sticky)
        return _W()

    def _f16_preflight(self, ro, rd, z, rd_view, expr, latent, near, far, max_rays=256, max_samples=8):
        """Range probe for the split-fp16 kernel (cf. ops.f16_preflight): the exact-f32 training forward on a strided sample of
        the chunk's points; returns the largest hidden |activation|."""
        n_rays, n_s = z.shape
        rs, ss = max(1, n_rays // max_rays), max(1, n_s // max_samples)
        rv = None if rd_view is None else rd_view[::rs].contiguous()
        keep = ops.get_mlp_precision()
        ops.set_mlp_precision("f32")
        try:
            with torch.enable_grad():
                _, state = self.hip_forward(ro[::rs].contiguous(), rd[::rs].contiguous(), z[::rs, ::ss].contiguous(), rv, expr, latent, near, far, True)
        finally:
            ops.set_mlp_precision(keep)
        saved = state[2]
        n = z[::rs, ::ss].numel()
        return float(saved[64 * n:1472 * n].abs().max().item())     # sections S_L1 .. S_DIR (csrc/nf_mlp_lcode_layout.h)

    def hip_forward(self, ro, rd, z, rd_view, expr, latent, near, far, need_grad):
        import numpy as np
        from . import _hip as H
        packed = self._hip_packed()
        lib = H.lib()
        dev = H.require_device(packed, ro, rd, z, rd_view, expr, latent)
        cond = torch.empty(lib.nf_lcode_cond_floats(), dtype=torch.float32, device=dev)
        n_rays, n_samples = z.shape
        raw = torch.empty((n_rays, n_samples, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            H.check(lib.nf_lcode_condition(H.ptr(packed), H.ptr(expr), H.ptr(latent), float(np.float32(near)), float(np.float32(far)),
                                           H.ptr(cond), H.stream_ptr(dev)), "nf_lcode_condition")
            if not need_grad:
                if ops.get_mlp_precision() == "bf16x3":
                    H.check(lib.nf_lcode_mlp_fwd_bf16(H.ptr(self._hip_packed_bf16()), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view),
                                                      H.ptr(z), n_rays, n_samples, H.ptr(raw), H.stream_ptr(dev)), "nf_lcode_mlp_fwd_bf16")
                elif ops.get_mlp_precision() in ops.F16_MODES:
                    key = (tuple((int(p.data_ptr()), int(p._version)) for p in self.hip_param_list()), expr.data_ptr(), latent.data_ptr(),
                           expr._version, latent._version)
                    if getattr(self, "_f16_probe_key", None) != key:      # once per (weights, conditioning): once per frame and model
                        amax = self._f16_preflight(ro, rd, z, rd_view, expr, latent, near, far)
                        if not amax * ops.F16_PREFLIGHT_MARGIN < ops.F16_ACT_LIMIT:
                            raise RuntimeError(f'nerf.set_mlp_precision("f16x3"): hidden activations of {type(self).__name__} reach {amax:.3g} '
                                               f'on a sample of this frame, within {ops.F16_PREFLIGHT_MARGIN:g}x of the fp16 range limit '
                                               f'({ops.F16_ACT_LIMIT:g}) -- render this model with "f32" or "bf16x3"')
                        self._f16_probe_key = key
                    fwd16 = lib.nf_lcode_mlp_fwd_f16x2 if ops.get_mlp_precision() == "f16x2" else lib.nf_lcode_mlp_fwd_f16
                    H.check(fwd16(H.ptr(self._hip_pack("f16")), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view),
                                  H.ptr(z), n_rays, n_samples, H.ptr(raw), H.stream_ptr(dev)), "nf_lcode_mlp_fwd_f16[x2]")
                else:
                    H.check(lib.nf_lcode_mlp_fwd(H.ptr(packed), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view), H.ptr(z), n_rays,
                                                 n_samples, H.ptr(raw), H.stream_ptr(dev)), "nf_lcode_mlp_fwd")
                return raw, None
            ops.require_trainable_precision()
            prec = ops.get_mlp_precision()
            split = "f16" if prec == "f16x3" else prec == "bf16x3"
            saved = torch.empty(lib.nf_lcode_saved_floats(n_rays * n_samples), dtype=torch.float32, device=dev)
            if split == "f16" and not getattr(self, "_in_f16_probe", False):
                # range probe (weights move every step): the first training call and every 128th
                n_calls = self.__dict__["_f16_train_calls"] = self.__dict__.get("_f16_train_calls", 0) + 1
                if ops.f16_train_probe_every() == 1 or n_calls % ops.f16_train_probe_every() == 1:
                    self._in_f16_probe = True
                    try:
                        amax = self._f16_preflight(ro, rd, z, rd_view, expr, latent, near, far)
                    finally:
                        self._in_f16_probe = False
                    if not amax * ops.F16_PREFLIGHT_MARGIN < ops.F16_ACT_LIMIT:
                        raise RuntimeError(f'nerf.set_mlp_precision("f16x3"): hidden activations of {type(self).__name__} reach {amax:.3g}, '
                                           f'within {ops.F16_PREFLIGHT_MARGIN:g}x of the fp16 range limit ({ops.F16_ACT_LIMIT:g}) -- train '
                                           f'this model with "f32" or "bf16x3"')
            if split == "f16":
                H.check(lib.nf_lcode_mlp_fwd_train_f16(H.ptr(self._hip_pack("f16")), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view),
                                                       H.ptr(z), n_rays, n_samples, H.ptr(raw), H.ptr(saved), H.stream_ptr(dev)),
                        "nf_lcode_mlp_fwd_train_f16")
            elif split:
                H.check(lib.nf_lcode_mlp_fwd_train_bf16(H.ptr(self._hip_packed_bf16()), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view),
                                                        H.ptr(z), n_rays, n_samples, H.ptr(raw), H.ptr(saved), H.stream_ptr(dev)),
                        "nf_lcode_mlp_fwd_train_bf16")
            else:
                H.check(lib.nf_lcode_mlp_fwd_train(H.ptr(packed), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd_view), H.ptr(z), n_rays,
                                                   n_samples, H.ptr(raw), H.ptr(saved), H.stream_ptr(dev)), "nf_lcode_mlp_fwd_train")
        return raw, (packed, cond, saved, split)

    def hip_backward(self, state, z, d_raw):
        """d_raw (n_rays, n_samples, 4) -> ([16 parameter gradients in hip_param_list order], d latent (32))."""
        from . import _hip as H
        packed, cond, saved, split = state
        lib = H.lib()
        d_raw = d_raw.contiguous()
        dev = H.require_device(packed, cond, saved, d_raw)
        n_rays, n_samples = z.shape
        with torch.cuda.device(dev):         # sized from the CURRENT device's CU count (nf_mlp_dw.h): ask on `dev`
            ws_floats = lib.nf_lcode_bwd_workspace_floats(n_rays * n_samples)
        ws = torch.empty(ws_floats, dtype=torch.float32, device=dev)
        flat = torch.empty(lib.nf_lcode_grad_floats(), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if split == "f16":     # forward was the split-fp16 training forward: split-fp16 chain + dW GEMMs
                H.check(lib.nf_lcode_mlp_bwd_f16(H.ptr(packed), H.ptr(self._hip_pack("f16_t")), H.ptr(cond), H.ptr(saved), H.ptr(d_raw),
                                                 n_rays, n_samples, H.ptr(ws), ws_floats, H.ptr(flat), H.stream_ptr(dev)),
                        "nf_lcode_mlp_bwd_f16")
            elif split:     # forward was the split-bf16 training forward (bit masks present): split-bf16 chain + dW GEMMs
                H.check(lib.nf_lcode_mlp_bwd_bf16(H.ptr(packed), H.ptr(self._hip_packed_bf16_t()), H.ptr(cond), H.ptr(saved), H.ptr(d_raw),
                                                  n_rays, n_samples, H.ptr(ws), ws_floats, H.ptr(flat), H.stream_ptr(dev)),
                        "nf_lcode_mlp_bwd_bf16")
            else:
                H.check(lib.nf_lcode_mlp_bwd(H.ptr(packed), H.ptr(self._hip_packed_t()), H.ptr(cond), H.ptr(saved), H.ptr(d_raw), n_rays,
                                             n_samples, H.ptr(ws), ws_floats, H.ptr(flat), H.stream_ptr(dev)), "nf_lcode_mlp_bwd")
        grads, off = [], 0
        for p in self.hip_param_list():
            n = p.numel()
            grads.append(flat[off:off + n].view(p.shape))
            off += n
        return grads, flat[off:off + 32]

    def forward(self, x, expr=None, latent_code=None, **kwargs):
        """M:590-636 on pre-encoded inputs x (N, 87) = [PE10(xyz) | PE4(dirs)] -> (N, 4), as run_network calls it (T:9-33).
        Inference only (kernel nf_lcode_forward_encoded); training goes through run_one_iter_of_nerf, whose fused kernels own
        the backward."""
        from . import _hip as H
        if not self.fused_supported() or expr is None or latent_code is None:
            raise NotImplementedError("forward() is built for the NeRFace geometry and needs expr and latent_code")
        if torch.is_grad_enabled() and (x.requires_grad or latent_code.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("ConditionalBlendshapeLearnableCodeNeRFModel.forward has no autograd on the MI355X build: train "
                                      "through nerf.run_one_iter_of_nerf(...), or call forward under torch.no_grad()")
        x = ops._c(x.detach())
        if x.dim() != 2 or x.shape[1] != 87:
            raise ValueError("expected pre-encoded inputs of shape (N, 87)")
        expr_d, lat_d = ops._c(expr.detach()).reshape(-1), ops._c(latent_code.detach()).reshape(-1)
        packed = self._hip_packed()
        dev = H.require_device(packed, x, expr_d, lat_d)
        lib = H.lib()
        cond = torch.empty(lib.nf_lcode_cond_floats(), dtype=torch.float32, device=dev)
        out = torch.empty((x.shape[0], 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            H.check(lib.nf_lcode_forward_encoded(H.ptr(packed), H.ptr(x), H.ptr(expr_d), H.ptr(lat_d), x.shape[0], H.ptr(cond),
                                                 H.ptr(out), H.stream_ptr(dev)), "nf_lcode_forward_encoded")
        return out


class FlexibleNeRFModel(torch.nn.Module):
    r"""Reference nerf/models.py:351-422: constructor signature, parameter names, shapes and default initialisation as there, so
    its state_dicts load unchanged.

    The HIP path is built for the configuration that reads BASELINE config 1 literally ("tiny_nerf ... 4-layer MLP"):
    `FlexibleNeRFModel(num_layers=L, hidden_size=128, num_encoding_fn_xyz=10, include_input_xyz=True, use_viewdirs=False)` with
    L = 2 .. 5 (no skip connection fires below 6 layers, M:373 / 404-409):
    PE10(xyz) (63) -> layer1 (Linear 128, NO activation, M:402) -> (L - 1) x (Linear 128 + ReLU) -> fc_out (Linear 4).
    It is evaluated inside the fused tiny-path kernels (`tiny_nerf.run_one_iter_of_tinynerf(..., model, 10)`: nf_flex_mlp_fwd and, with
    gradients enabled, nf_flex_mlp_fwd_train / nf_flex_mlp_bwd); other geometries construct and load but have no kernel and raise."""

    def __init__(self, num_layers=4, hidden_size=128, skip_connect_every=4, num_encoding_fn_xyz=6, num_encoding_fn_dir=4,
                 include_input_xyz=True, include_input_dir=True, use_viewdirs=True):
        super().__init__()
        include_input_xyz = 3 if include_input_xyz else 0
        include_input_dir = 3 if include_input_dir else 0
        self.dim_xyz = include_input_xyz + 2 * 3 * num_encoding_fn_xyz
        self.dim_dir = include_input_dir + 2 * 3 * num_encoding_fn_dir
        self.skip_connect_every = skip_connect_every
        if not use_viewdirs:
            self.dim_dir = 0
        self.layer1 = torch.nn.Linear(self.dim_xyz, hidden_size)
        self.layers_xyz = torch.nn.ModuleList()
        for i in range(num_layers - 1):
            skip = i % self.skip_connect_every == 0 and i > 0 and i != num_layers - 1                                  # M:373
            self.layers_xyz.append(torch.nn.Linear(self.dim_xyz + hidden_size if skip else hidden_size, hidden_size))
        self.use_viewdirs = use_viewdirs
        if self.use_viewdirs:
            self.layers_dir = torch.nn.ModuleList()
            self.layers_dir.append(torch.nn.Linear(self.dim_dir + hidden_size, hidden_size // 2))
            self.fc_alpha = torch.nn.Linear(hidden_size, 1)
            self.fc_rgb = torch.nn.Linear(hidden_size // 2, 3)
            self.fc_feat = torch.nn.Linear(hidden_size, hidden_size)
        else:
            self.fc_out = torch.nn.Linear(hidden_size, 4)
        self.relu = torch.nn.functional.relu

    @property
    def num_layers(self) -> int:
        return len(self.layers_xyz) + 1

    def fused_supported(self) -> bool:
        return (not self.use_viewdirs and self.dim_xyz == 63 and self.layer1.out_features == 128 and 2 <= self.num_layers <= 5
                and all(l.in_features == 128 for l in self.layers_xyz))

    def hip_param_list(self):
        """state_dict order: layer1, layers_xyz.0 .. , fc_out (weight, bias each) -- the order of nf_flex_pack / nf_flex_mlp_bwd."""
        ps = [self.layer1.weight, self.layer1.bias]
        for l in self.layers_xyz:
            ps += [l.weight, l.bias]
        return ps + [self.fc_out.weight, self.fc_out.bias]

    def forward(self, x):
        raise NotImplementedError("FlexibleNeRFModel is evaluated inside the fused tiny-path kernel: call "
                                  "tiny_nerf.run_one_iter_of_tinynerf(..., model, 10)")
