"""`nerf.optim.Adam` -- the trainer's optimizer step as ONE kernel launch (nf_adam_step, csrc/nf_optim.hip).

The reference steps torch.optim.Adam over [coarse model, fine model, latent codes] (train_transformed_rays.py:193-199, 391-392).
torch's multi-tensor paths spend ~110 us per step on the 54 tensors of that list (two multi_tensor_apply kernels, one of them only to
increment 54 step counters); this class runs torch's update rule for every tensor of a parameter group in one launch (~8 us).

Drop-in for torch.optim.Adam where the trainer uses it: same constructor arguments, same `param_groups`, and the SAME state layout
(`state[p] = {"step": 0-d float32 CPU tensor, "exp_avg", "exp_avg_sq"}`), so `state_dict()` / `load_state_dict()` round-trip with
torch.optim.Adam and with the reference's checkpoints.  Supported: fp32 parameters on a ROCm device, weight_decay = 0, amsgrad =
False, maximize = False -- what the reference uses; anything else raises (no fallback).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _hip as H


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False, **unused):
        if weight_decay != 0 or amsgrad or maximize:
            raise NotImplementedError("nerf.optim.Adam implements the trainer's configuration: weight_decay=0, amsgrad=False, maximize=False")
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=0, amsgrad=False, maximize=False))

    def __setstate__(self, state):
        """torch.optim.Adam.__setstate__'s normalisation: checkpoints written by the reference's torch versions (<= 1.11) hold
        `step` as a Python int, fused / capturable ones as a device tensor; here it is always a 0-d float32 CPU tensor."""
        super().__setstate__(state)
        for st in self.state.values():
            if "step" in st:
                s_ = st["step"]
                st["step"] = torch.tensor(float(s_.item() if torch.is_tensor(s_) else s_), dtype=torch.float32)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = H.lib()
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("nerf.optim.Adam: weight_decay / amsgrad / maximize are not part of the trainer's configuration")
            todo = []
            for p in group["params"]:                      # validate EVERYTHING first: a refused tensor must not leave others half-stepped
                if p.grad is None or p.numel() == 0:       # (zero-element parameters: nothing to update, as in torch.optim.Adam)
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("nerf.optim.Adam does not support sparse gradients")
                g = p.grad
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("nerf.optim.Adam (MI355X build): parameters must be contiguous float32 tensors on a ROCm device")
                if g.dtype != torch.float32 or g.device != p.device:
                    raise RuntimeError("nerf.optim.Adam: gradients must be float32 on the parameter's device")
                todo.append((p, g if g.is_contiguous() else g.contiguous()))
            by_step = {}                                   # tensors that have taken the same number of steps go into one launch
            for p, g in todo:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif not torch.is_tensor(st["step"]) or st["step"].is_cuda:      # int (old checkpoints) / device tensor: normalise once
                    s_ = st["step"]
                    st["step"] = torch.tensor(float(s_.item() if torch.is_tensor(s_) else s_), dtype=torch.float32)
                st["step"] += 1
                by_step.setdefault((int(st["step"].item()), p.device), []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            beta1, beta2 = group["betas"]
            for (step, dev), items in by_step.items():
                n = len(items)
                arr = lambda k: (C.c_void_p * n)(*[int(it[k].data_ptr()) for it in items])
                numel = (C.c_int64 * n)(*[int(it[0].numel()) for it in items])
                with torch.cuda.device(dev):
                    H.check(lib.nf_adam_step(arr(0), arr(1), arr(2), arr(3), numel, n, float(group["lr"]), float(beta1), float(beta2),
                                             float(group["eps"]), step, H.stream_ptr(dev)), "nf_adam_step")
                # the kernel wrote through raw pointers: tell autograd (in-place checks, and every cache keyed on `_version`:
                # the split-fp16 range probe of nerf/models.py) that these tensors changed
                for it in items:
                    torch.autograd.graph.increment_version(it[0])
        return loss
