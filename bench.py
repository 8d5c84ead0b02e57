#!/usr/bin/env python
"""Benchmark of the NeRFace ray-marching hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One *step* = one full 512x512 frame through the product path exactly as eval_transformed_rays.py drives it:
`get_ray_bundle` + `run_one_iter_of_nerf(mode="validation")` with the shipped validation settings (64 coarse + 128 fine
samples, chunksize 65536, perturb: True, noise 0; paper model x2, expression + latent conditioning, background prior).
Inputs (pose, expression, latent code, background, weights) are resident in HBM before the timed region.
Metric (BASELINE.json): rays/sec = frames * 262144 / wall time, whole job.  The headline (`value`, `dtype` "f32") runs the
exact-f32 MFMA kernels -- the reference's arithmetic (fp32 everywhere, nerf/models.py:236-261).

N > 1: frames are sharded over ranks (eval is embarrassingly parallel); no data-path collective; each rank
renders K frames of its own (weak scaling); time = max over ranks.

On the same JSON line (every BASELINE config that fits this box is timed by this one command):
  roofline     -- the dominant kernel (fused MLP forward, fine pass: 65536 rays x 192 samples per launch), timed live with
                  HIP events on the launch stream; achieved = algorithmic FLOPs (1,100,032 per point) / average
                  launch duration, against the dense fp32-MFMA peak (157.3 TFLOP/s); `traffic` = HBM bytes per launch from
                  rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes) run by this command on the same kernel.
  split_bf16   -- the same K frames with nerf.set_mlp_precision("bf16x3") (three bf16 MFMAs per product, f32 accumulate):
                  value, ms, its kernel's roofline against the dense bf16 peak, and its parity on the CPU sample.
  train        -- configs[2]: 2048-ray training iterations (64+64, fwd + bwd + Adam) in the three arithmetics, ms/iter, rays/s and
                  a per-kernel roofline (forward with saves, dX chain, weight-gradient GEMMs: exact f32 against the fp32-MFMA
                  peak, the split arithmetics against HBM; PMC traffic per launch) (N>1: configs[4], data parallel, flat all-reduce).
  tiny         -- configs[0]: tiny_nerf 64x64x32 forward on the device next to the CPU oracle of the same image.
  eager_rocm   -- the reference algorithm as stock PyTorch-ROCm eager fp32 ops on the same GPU (bounded ray sample): the same-box
                  denominator BASELINE.md names next to the CPU one.
  cpu_baseline -- kind "reference": the UNMODIFIED reference's get_ray_bundle + run_one_iter_of_nerf (imported out of /root/reference, or
                  out of oracle/_ref/nerface_ref.zip which oracle/make_ref.py packs and which travels with the push) timed on this box's
                  host cores on a bounded sample of the same workload (rank 0, N=1 only), the oracle port beside it on a slice;
                  kind "port" (labelled fallback) only where neither is present.
  summary      -- LAST key of the line: flat scalars (the driver's record keeps only the tail of the line).

Files: bench.py (this: the timed regions, the training object, the line), bench_common.py (workload constants, synthetic scene),
bench_baselines.py (cpu_baseline / eager_rocm legs -- the only importers of oracle/), bench_probes.py (device, power, PMC, launcher legs).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import time

# the host driver of this pool only supports dmabuf IPC: without this RCCL's peer mappings fail at N > 1 (exported by the image
# already; set here too so that a bare `python -m torch.distributed.run ... bench.py` from a clean shell works)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "4d-facial-avatars_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# the workload's constants and synthetic scene (bench.INTRINSICS, bench.synth_params, ... stay importable from here), the baseline legs
# (the only importers of oracle/ outside tests/) and the probe legs (device / power / PMC / launcher): VERDICT r05 housekeeping
from bench_common import *  # noqa: E402,F401,F403
from bench_baselines import cpu_baseline, cpu_baseline_tiny, eager_rocm_baseline, eager_rocm_reference  # noqa: E402,F401
from bench_probes import (_pmc_guard, device_info, launcher_eval_leg, pattern_store_probe, pmc_kernel_bytes, pmc_pass_rows,  # noqa: E402,F401
                          pmc_sustained_clock, pmc_traffic, pmc_train_clocks, pmc_train_traffic, power_probe)


def train_roofline(args, model, dev, n_rays):
    """Per-kernel roofline of the training MLP kernels of ONE model, measured live with HIP events on the stream they are launched
    on: the training forward (its own C call) and the three stages of the backward call (nf_paper_mlp_bwd_stage_ms records events
    between the dX chain, the weight-gradient GEMMs and the slab reduction), at the two launch sizes of an iteration
    (n_rays x 64 coarse, n_rays x 128 fine).  Each kernel is priced against the bound that binds it: exact f32 -> the dense
    fp32-MFMA peak (it issues one MFMA FLOP per algorithmic FLOP and moves 9-18 KB per point in the same time: MFMA-bound);
    split arithmetics (3 x 16-bit MFMAs per product, 16x the rate) -> HBM."""
    import ctypes as C
    from nerf import _hip as HH
    from nerf import ops
    prec = args.precision
    lib = HH.lib()
    g = torch.Generator(device="cpu").manual_seed(5)
    ro = torch.zeros(n_rays, 3).to(dev)
    rd = (torch.randn(n_rays, 3, generator=g) * 0.3).to(dev)
    expr, lat = (torch.randn(76, generator=g) * 0.5).to(dev), (torch.randn(32, generator=g) * 0.1).to(dev)
    if args.family != "paper":                                             # second family: the whole-iteration figure only
        total_ms, total_pts = 0.0, 0
        for s_ in (64, 128):
            z = torch.sort(torch.rand(n_rays, s_, generator=g) * 0.6 + 0.2, dim=-1)[0].to(dev)
            d_raw = (torch.randn(n_rays, s_, 4, generator=g) / (3 * n_rays)).to(dev)

            def once():
                raw, state = model.hip_forward(ro, rd, z, rd, expr, lat, NEAR, FAR, True)
                model.hip_backward(state, z, d_raw)
            for _ in range(3):
                once()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                once()
            e1.record()
            torch.cuda.synchronize()
            total_ms += e0.elapsed_time(e1) / 10
            total_pts += n_rays * s_
        bpp = LCODE_BYTES_PER_POINT[prec]
        ach = bpp * total_pts / (total_ms * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "lcode MLP training kernels of ONE model per iteration", "achieved": ach, "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None, "algorithmic_bytes_per_point": bpp, "ms_both_launches": total_ms}
    hw = model.hip_weights()
    packed = hw.get()
    cond = ops.paper_condition(packed, expr, lat, NEAR, FAR)
    pk_fwd = {"f32": None, "bf16x3": hw.get_bf16(), "f16x3": hw.get_f16()}[prec]
    pk_bwd = {"f32": hw.get_t, "bf16x3": hw.get_bf16_t, "f16x3": hw.get_f16_t}[prec]()
    code = {"f32": 0, "bf16x3": 1, "f16x3": 2}[prec]
    flat = torch.empty(lib.nf_paper_grad_floats(), device=dev)
    per_size = {}
    reps = 8
    for s_ in (64, 128):
        n = n_rays * s_
        z = torch.sort(torch.rand(n_rays, s_, generator=g) * 0.6 + 0.2, dim=-1)[0].to(dev)
        d_raw = (torch.randn(n_rays, s_, 4, generator=g) / (3 * n_rays)).to(dev)
        ws_floats = lib.nf_paper_bwd_workspace_floats(n)
        ws = torch.empty(ws_floats, device=dev)
        ms = [0.0, 0.0, 0.0, 0.0]
        fwd = lambda: ops.paper_mlp_fwd_train(packed, cond, ro, rd, z, rd, packed_b=pk_fwd if prec == "bf16x3" else None,
                                              packed_h=pk_fwd if prec == "f16x3" else None)
        for it in range(reps + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            # the timed launch queues behind an untimed one of the same kernel: round 5 recorded e0 on an IDLE stream (the backward call
            # below synchronises), so e0 -> e1 also held the host's 30-60 us between the event and the launch (two torch.empty, the ctypes
            # call) and the clock ramp of a GPU that had gone idle -- 2-4 % of a 1-2 ms kernel that runs behind other kernels in a real
            # iteration.  The backward stages were always timed by events recorded inside ONE C call, back to back on a busy queue.
            del_me = fwd()
            e0.record()
            raw, (saved,) = fwd()
            e1.record()
            del del_me
            st = (C.c_float * 3)()
            HH.check(lib.nf_paper_mlp_bwd_stage_ms(HH.ptr(packed), HH.ptr(pk_bwd), code, HH.ptr(cond), HH.ptr(saved), HH.ptr(d_raw), n_rays, s_,
                                                   HH.ptr(ws), ws_floats, HH.ptr(flat), st, HH.stream_ptr(dev)), "nf_paper_mlp_bwd_stage_ms")
            if it >= 2:                                                    # (the call above synchronised the stream)
                ms[0] += e0.elapsed_time(e1) / reps
                for k in range(3):
                    ms[k + 1] += st[k] / reps
            del saved
        per_size[s_] = ms
        del ws
    n_big = n_rays * 128
    kernels = []
    for k, (kname, what) in enumerate(TRAIN_KERNELS[prec]):
        t = per_size[128][k] * 1e-3
        if prec == "f32":
            ach = TRAIN_FLOP_PER_POINT[k] * n_big / t / 1e12
            obj = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                   "algorithmic_flops_per_point": TRAIN_FLOP_PER_POINT[k]}
            if k == 0:                                                   # frac = the executed (physical) fraction, like the headline's
                obj["executed_tflops"] = EXEC_FLOP_PER_POINT_F32 * n_big / t / 1e12
                obj["frac_executed"] = obj["executed_tflops"] / PEAK_F32_MFMA_TFLOPS
                obj["frac_algorithmic"], obj["frac"] = obj["frac"], obj["frac_executed"]
        else:
            # priced against HBM (9-18 KB per point: the bound the bytes set) AND, beside it, against what the matrix pipe could do: the
            # issued 16-bit MFMA FLOPs of the launch / time / the dense 16-bit peak (these kernels are power- /
            # issue-bound, not HBM-bound -- the HBM fraction alone hides how far they sit from the matrix roofline)
            ach = TRAIN_BYTES_PER_POINT[k] * n_big / t / 1e9
            exe = TRAIN_SPLIT_EXEC_FLOP_PER_POINT[k] * n_big / t / 1e12
            obj = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
                   "executed_mfma_tflops": exe, "frac_executed_mfma": exe / PEAK_BF16_MFMA_TFLOPS,
                   "executed_mfma_flops_per_point": TRAIN_SPLIT_EXEC_FLOP_PER_POINT[k]}
        obj.update({"kernel": f"{kname} ({what}; {n_rays} rays x 128 samples per launch)", "avg_launch_ms": per_size[128][k],
                    "avg_launch_ms_64_samples": per_size[64][k], "algorithmic_hbm_bytes_per_point": TRAIN_BYTES_PER_POINT[k], "traffic": None})
        kernels.append(obj)
    total = sum(per_size[64]) + sum(per_size[128])
    top = max(kernels, key=lambda o: o["avg_launch_ms"])
    return {**{k: top[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_launch_ms", "traffic")},   # the dominant kernel
            "kernels": kernels, "reduce_unpack_ms": {"64_samples": per_size[64][3], "128_samples": per_size[128][3]},
            "ms_both_launches": total,
            "note": "per iteration the coarse model runs the 64-sample launch and the fine model the 128-sample launch, so ms_both_launches "
                    "is the MLP-kernel time of one training iteration; forward timed around its own call, backward stages by HIP events "
                    "recorded inside nf_paper_mlp_bwd_stage_ms on the launch stream; `traffic` = PMC HBM bytes per launch (filled by the "
                    "eval line's PMC passes, tools/pmc_train_launch.py)"}


def bench_train(args, nerf, model_c, model_f, dev, rank, world, dist, emit=True):
    """configs[2] / configs[4]: the trainer's iteration (TR:289-400) on synthetic data -- full-frame ray bundle, 2048 random
    rays, run_one_iter_of_nerf(mode='train') with the shipped training settings (64+64, chunksize 2048, perturb, noise
    0.1), coarse+fine MSE + latent regulariser, backward, (N>1: one flat RCCL all-reduce), Adam over
    [coarse, fine, latent table]."""
    from nerf import distributed as D
    n_rays, n_train = 2048, 1000
    mode = dict(num_coarse=64, num_fine=64, chunksize=2048, perturb=True, lindisp=False, radiance_field_noise_std=0.1,
                white_background=False)
    opt = nerf.CfgNode(dict(nerf=dict(use_viewdirs=True, train=dict(mode), validation=dict(mode)),
                            dataset=dict(no_ndc=True, near=NEAR, far=FAR)))
    enc_xyz = nerf.get_embedding_function(num_encoding_functions=10, include_input=True, log_sampling=True)
    enc_dir = nerf.get_embedding_function(num_encoding_functions=4, include_input=False, log_sampling=True)
    model_c.train()
    model_f.train()
    latent_codes = torch.zeros(n_train, 32, device=dev).requires_grad_(True)
    params = list(model_c.parameters()) + list(model_f.parameters()) + [latent_codes]
    D.broadcast_parameters(params)
    optim = nerf.optim.Adam(params, lr=5e-4)                   # torch.optim.Adam's update rule (TR:193-199) for all 54 tensors in one launch
    reducer = D.GradientAllReducer(params)
    reducer.enable_timing()                                               # HIP events around the flat all-reduce (no-op at world 1 without a group)
    g = torch.Generator().manual_seed(7)
    background = torch.rand((H, W, 3), generator=g).to(dev)
    target = torch.rand((H, W, 3), generator=g).to(dev)
    torch.manual_seed(D.rank_seed(1234))
    importance = torch.full((H, W), 0.1, device=dev)                   # the trainer's importance map (TR:230-239) for a centred face box
    importance[H // 5: 4 * H // 5, W // 4: 3 * W // 4] = 0.9
    importance = (importance / importance.sum()).reshape(-1)
    importance = importance.reshape(W, H).T.contiguous().reshape(-1)   # the reference applies the map transposed (launch.train_sharded.importance_maps)
    n_it = args.steps + args.warmup
    frame_ids = torch.randint(0, n_train, (n_it,)).tolist()
    poses = [frame_pose(f).to(dev) for f in frame_ids]
    exprs = [(0.5 * torch.randn(76)).to(dev) for _ in frame_ids]

    def step(i):
        # TR:320-322 on the device: 2048 distinct pixels, importance-sampled (p = 0.9 inside the face box, TR:230-239), then rays +
        # target pixels + background prior of the selected pixels in one kernel (the launcher's form of TR:302, 325-330)
        sel = nerf.choose_rays(importance, n_rays)
        ro, rd, tgt, bg = nerf.get_ray_batch(H, W, INTRINSICS, poses[i], sel, target, background)
        latent = latent_codes[frame_ids[i]]
        out = nerf.run_one_iter_of_nerf(H, W, INTRINSICS, model_c, model_f, ro, rd, opt,
                                        mode="train", encode_position_fn=enc_xyz, encode_direction_fn=enc_dir,
                                        expressions=exprs[i], background_prior=bg, latent_code=latent)
        loss, _ = nerf.training_loss(out[0], out[3], tgt, latent)      # TR:355-387 as the launcher computes it (two launches, fwd + bwd)
        loss.backward()
        reducer.reduce()
        optim.step()
        optim.zero_grad()
        return loss

    for i in range(args.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_it):
        loss = step(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(loss))
    roofline = train_roofline(args, model_f, dev, n_rays) if rank == 0 else None
    allreduce = reducer.stats()          # ranks the communicator saw, bytes per step, median HIP-event time of the collective
    result = None
    if rank == 0:
        result = {
            "roofline": roofline, "allreduce": allreduce, "ranks_seen": allreduce["ranks_seen"],
            "allreduce_us": allreduce["allreduce_us"], "bytes_allreduced": allreduce["bytes_allreduced"],
            "metric": "training rays/sec (2048 rays/iter, 64+64 samples, fwd+bwd+Adam)", "value": world * args.steps * n_rays / dt,
            "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16x3": "bf16x3 (split-bf16 products, f32 accumulate)", "f16x3": "f16x3 (split-fp16 products, f32 accumulate)"}.get(args.precision, "f32"),
            "data": "synthetic",
            "config": {"workload": f"configs[2]: {args.family}-model training iteration, 2048 rays from a 512x512 frame, 64+64 samples, "
                                   "noise 0.1, latent table 1000x32, Adam; one frame per rank, flat grad all-reduce",
                       "rays_per_step": n_rays * world, "parallelism": f"dp{world}",
                       "mlp_precision": args.precision, "family": args.family}}
        if emit:
            result["config"]["device"] = device_info(dev)
            result["summary"] = {"value_rays_s": result["value"], "ms_per_step": result["ms_per_step"], "n_gpus": world,
                                 "ranks_seen": allreduce["ranks_seen"], "allreduce_us": allreduce["allreduce_us"],
                                 "bytes_allreduced": allreduce["bytes_allreduced"], "mlp_precision": args.precision,
                                 "hbm_fill_gbs": result["config"]["device"].get("hbm_fill_gbs")}
            globals()["emit"](result)
    if dist is not None:
        dist.barrier()
        if emit:
            dist.destroy_process_group()
    return result


def bench_tiny(dev, steps=20, flex_layers=0):
    """configs[0]: tiny_nerf 64x64 image, 32 samples (TN:111-159) -- the fused tiny kernels on the device.  Returns the result
    and the inputs (weights, pose, focal) so that cpu_baseline_tiny can time the CPU oracle on the same image.
    flex_layers = L > 0: BASELINE's "4-layer MLP" read literally -- the reference's FlexibleNeRFModel(num_layers=L, 128, use_viewdirs=False)
    (M:351-422) in place of the script's own 3-Linear VeryTinyNerfModel (nf_flex_* kernels)."""
    import tiny_nerf as TN
    import nerf
    torch.cuda.empty_cache()                                            # the eval / training legs leave GBs of cached blocks behind
    torch.manual_seed(9458)                                             # TN:264

    def new_model():
        if flex_layers:
            return nerf.models.FlexibleNeRFModel(num_layers=flex_layers, hidden_size=128, num_encoding_fn_xyz=10, include_input_xyz=True,
                                                 use_viewdirs=False).to(dev)
        return TN.VeryTinyNerfModel(num_encoding_functions=10).to(dev)
    model = new_model()
    pose = frame_pose(7)
    pose[:3, 3] = torch.tensor([0.3, -0.2, 4.0])
    focal = torch.tensor(138.88 * 64 / 100.0)
    pose_d = pose.to(dev)

    def once():
        with torch.no_grad():
            return TN.run_one_iter_of_tinynerf(64, 64, focal, pose_d, 2.0, 6.0, 32, None, TN.get_minibatches, 16384, model, 10)
    for _ in range(3):
        rgb = once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rgb = once()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    assert rgb.shape == (64, 64, 3) and bool(torch.isfinite(rgb).all())
    # the reference script is a trainer (TN:282-302): forward + mse + backward + Adam on the same image
    target = torch.rand((64, 64, 3), generator=torch.Generator().manual_seed(13)).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)

    def train_once():
        rgb = TN.run_one_iter_of_tinynerf(64, 64, focal, pose_d, 2.0, 6.0, 32, None, TN.get_minibatches, 16384, model, 10)
        loss = torch.nn.functional.mse_loss(rgb, target)
        loss.backward()
        opt.step()
        opt.zero_grad()
        return loss
    for _ in range(5):
        train_once()
    ms_train = float("inf")
    for _ in range(3):                                                  # best of three batches: after the big frames of the eval
        torch.cuda.synchronize()                                        # legs the caching allocator may still be re-shaping its pools
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = train_once()
        torch.cuda.synchronize()
        ms_train = min(ms_train, 1e3 * (time.perf_counter() - t0) / steps)
    assert bool(torch.isfinite(loss))
    # the same loop body captured once in a HIP graph and replayed (tiny_nerf.GraphedTinyTrainer; jitter drawn on the device)
    model_g = new_model()
    model_g.load_state_dict(model.state_dict())
    trainer = TN.GraphedTinyTrainer(model_g, torch.optim.Adam(model_g.parameters(), lr=5e-3, capturable=True), 64, 64, focal, 2.0, 6.0, 32, dev)
    for _ in range(3):
        trainer.step(pose_d, target)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5 * steps):
        loss_g = trainer.step(pose_d, target)
    torch.cuda.synchronize()
    ms_graph = 1e3 * (time.perf_counter() - t0) / (5 * steps)
    assert bool(torch.isfinite(loss_g))
    arch = (f"FlexibleNeRFModel(num_layers={flex_layers}) (63-128" + "-128" * (flex_layers - 1) + "-4, first layer linear)") if flex_layers \
        else "VeryTinyNerfModel (63-128-128-4)"
    res = {"workload": f"configs[0]: tiny_nerf 64x64 image, 32 samples per ray, {arch}, forward",
           "value": 4096 / (ms * 1e-3), "unit": "rays/s", "ms_per_image": ms, "images": steps,
           "train": {"ms_per_iter": ms_train, "value": 4096 / (ms_train * 1e-3), "unit": "rays/s",
                     "what": "forward + mse + backward (HIP kernels) + Adam per 64x64 image (TN:282-302)",
                     "hip_graph": {"ms_per_iter": ms_graph, "value": 4096 / (ms_graph * 1e-3), "unit": "rays/s",
                                   "what": "the same iteration captured once in a HIP graph and replayed (GraphedTinyTrainer)"}},
           "note": "host-launch bound on the device (ray bundle + two kernels per image of 4096 rays)"}
    return res, ({k: v.detach().cpu() for k, v in model.state_dict().items()}, pose, focal)


def self_launch(n):
    """Re-exec this command under torch.distributed.run with n ranks on this node; returns the launcher's exit code."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only (skip split_bf16 / train / tiny / PMC traffic)")
    ap.add_argument("--precision", choices=["bf16x3", "f16x3", "f16x2", "f32"], default="f32",
                    help="arithmetic of the HEADLINE: f32 (default) = exact-f32 MFMA, the reference's arithmetic; f16x3 = split-fp16 "
                         "(3 fp16 MFMAs per product on scaled weights, f32 accumulate: error against fp64 at the exact-f32 kernel's "
                         "level, the 1e-4 dB PSNR gate held against realistic targets); bf16x3 = split-bf16 and f16x2 = two fp16 products "
                         "(faster; they hold the gate against a uniform-random target, not against a 30 dB one on a sharp-density scene: "
                         "profiles/r06_gate_sensitivity.md).  The others are reported beside it (`exact_f32` / `split_f16` / `split_bf16` / `split_f16x2`)")
    ap.add_argument("--chunksize", type=int, default=CHUNK, help="validation ray chunk (shipped configs: 65536)")
    ap.add_argument("--family", choices=["paper", "lcode"], default="paper",
                    help="train mode only: lcode = ConditionalBlendshapeLearnableCodeNeRFModel")
    ap.add_argument("--mode", choices=["eval", "train"], default="eval",
                    help="eval (default) = BASELINE.json's metric; train = configs[2]/[4] as the headline: 2048 rays/iter, 64+64, fwd+bwd+Adam")
    ap.add_argument("--cpu-rays", type=int, default=12288)
    ap.add_argument("--train-steps", type=int, default=40, help="iterations of the `train` object of the eval line")
    ap.add_argument("--launcher-frames", type=int, default=32, help="frames of the launch/eval_sharded.py throughput leg")
    args = ap.parse_args()

    backend = os.environ.get("NERFACE_DIST_BACKEND", "nccl")           # "gloo": several ranks on one GPU (tests of the N > 1 path)
    if args.gpus > 1 and backend == "nccl" and torch.cuda.is_available() and args.gpus > torch.cuda.device_count():
        # fail in seconds with the reason, before any rendezvous: RCCL needs one device per rank (two ranks on one device deadlock
        # or abort inside ncclCommInitRank minutes later)
        raise SystemExit(f"bench.py: --gpus {args.gpus} needs {args.gpus} devices for the nccl (RCCL) backend, but "
                         f"torch.cuda.device_count() = {torch.cuda.device_count()} on this box (NERFACE_DIST_BACKEND=gloo runs several "
                         "ranks on one GPU for tests)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # a bare `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1) with the
        # same argv; the children see WORLD_SIZE and take the branch below.  --gpus 1 stays in-process.
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree "
                         "(the line reports n_gpus = ranks that ran)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the product has no CPU path)")
    if backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_pg = os.environ.get("NERFACE_DIST_FORCE", "0") not in ("", "0")   # world 1 under torch.distributed.run: run the collectives anyway (RCCL smoke)
    # the library BEFORE the process group: a cold box may have to compile it (minutes), and a rank that compiles while the others
    # sit in RCCL's init / first barrier can run them into the collective timeout.  Every rank takes a file lock; the first builds
    # (a no-op when the pushed .so is fresh), the others find it fresh.
    import fcntl
    import __graft_entry__ as G
    os.makedirs(os.path.join(PKG, "lib"), exist_ok=True)
    with open(os.path.join(PKG, "lib", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            G._load_build().build(force=False, verbose=False)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if world > 1 or force_pg:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(minutes=10))   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend=backend, timeout=datetime.timedelta(minutes=10))
    import nerf
    from nerf import ops
    nerf.set_mlp_precision(args.precision)

    if args.mode == "train" and args.precision in ops.INFERENCE_ONLY_PRECISIONS:
        raise SystemExit(f"--mode train: {args.precision} is an inference arithmetic (train with f32, f16x3 or bf16x3)")
    if args.family != "paper" and args.mode != "train":
        raise SystemExit("--family lcode is a --mode train option (the eval line is BASELINE.json's paper-model metric)")
    model_c, model_f = synth_params(0, dev, args.family), synth_params(1, dev, args.family)
    if args.mode == "train":
        return bench_train(args, nerf, model_c, model_f, dev, rank, world, dist)
    opt = options(nerf, args.chunksize)
    enc_xyz = nerf.get_embedding_function(num_encoding_functions=10, include_input=True, log_sampling=True)
    enc_dir = nerf.get_embedding_function(num_encoding_functions=4, include_input=False, log_sampling=True)
    g = torch.Generator().manual_seed(7)
    background = torch.rand((H, W, 3), generator=g).to(dev).view(-1, 3)
    n_frames = args.steps + args.warmup
    frames = [rank + world * i for i in range(n_frames)]                  # frame-parallel shard: f = rank (mod world)
    poses = [frame_pose(f).to(dev) for f in frames]
    conds = []
    for f in frames:
        gg = torch.Generator().manual_seed(1000 + f)
        conds.append(((0.5 * torch.randn(76, generator=gg)).to(dev), (0.1 * torch.randn(32, generator=gg)).to(dev)))
    torch.manual_seed(1234 + rank)

    def step(i):
        with torch.no_grad():
            ro, rd = nerf.get_ray_bundle(H, W, INTRINSICS, poses[i])
            return nerf.run_one_iter_of_nerf(H, W, INTRINSICS, model_c, model_f, ro, rd, opt, mode="validation",
                                             encode_position_fn=enc_xyz, encode_direction_fn=enc_dir,
                                             expressions=conds[i][0], background_prior=background, latent_code=conds[i][1])

    def timed_frames():
        """W warm-up frames, then exactly K frames between barrier + synchronize on both sides.  Returns (seconds of the slowest rank,
        [ms per step of every rank])."""
        for i in range(args.warmup):
            out = step(i)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.warmup, n_frames):
            out = step(i)
        torch.cuda.synchronize()
        own = time.perf_counter() - t0                                       # this rank's own K frames, before it waits for the others
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt, own], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")   # (gloo gathers host tensors only)
            per = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(per, t)
            per_rank = [1e3 * float(x[1].item()) / max(args.steps, 1) for x in per]    # every rank's own time: stragglers show on the line
            dt = max(float(x[0].item()) for x in per)                        # the job's time = the slowest rank's, closing barrier included
        else:
            per_rank = [1e3 * own / max(args.steps, 1)]
        assert out[3].shape == (H, W, 3) and bool(torch.isfinite(out[3]).all())
        return dt, per_rank

    dt, per_rank_ms = timed_frames()
    rays_total = world * args.steps * H * W
    # what each arithmetic is, and which targets it keeps north_star's 1e-4 dB gate against (profiles/r06_gate_sensitivity.md, nerf.gate.EXPECTED_PASS)
    dtype_of = {"f32": "f32",
                "bf16x3": "bf16x3 (split-bf16 products, f32 accumulate; gate: whole frames vs a uniform-random or 20 dB target, marginal at 30 dB on a x1000 density head)",
                "f16x3": "f16x3 (split-fp16 products on scaled weights, f32 accumulate; fp32-class: gate held on whole frames vs every target tried, 20-40 dB included)",
                "f16x2": "f16x2 (two fp16 products per weight: 11-bit activations, 22-bit weights, f32 accumulate; inference only; gate: whole frames vs a "
                         "uniform-random target only on a x1000 density head -- a speed arithmetic, not a matched-PSNR one)"}
    key_of = {"f32": "exact_f32", "bf16x3": "split_bf16", "f16x3": "split_f16", "f16x2": "split_f16x2"}
    line = {
        "metric": "rays/sec at 512x512, 64 coarse + 128 fine samples", "value": rays_total / dt, "unit": "rays/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype_of[args.precision], "data": "synthetic",
        "config": {"workload": "configs[1]: paper-model eval forward, 512x512 frame, 64+128 samples, chunksize 65536, "
                               "perturb on, expression+latent conditioned, background prior; frames sharded over GPUs",
                   "rays_per_step": H * W, "points_per_ray": N_COARSE + N_COARSE + N_FINE, "parallelism": f"frames x{world}",
                   "mlp_precision": args.precision, "device": device_info(dev)},
    }

    # ---- the same K frames (same warm-up, same bracketing) in the other arithmetics, beside the headline -----------------
    others = [p for p in ("f32", "f16x3", "f16x2", "bf16x3") if p != args.precision]
    if not args.no_extras:
        for other in others:
            nerf.set_mlp_precision(other)
            dt_o, _ = timed_frames()
            line[key_of[other]] = {
                "value": rays_total / dt_o, "unit": "rays/s", "ms_per_step": 1e3 * dt_o / max(args.steps, 1), "steps": args.steps,
                "warmup": args.warmup, "dtype": dtype_of[other],
                "note": f"same workload, frames and timing protocol with nerf.set_mlp_precision('{other}')"}
        nerf.set_mlp_precision(args.precision)
        if rank == 0 and args.precision == "f32":
            # north_star's gate on the workload itself, where it is hard (nerf/gate.py): the first four frames of the timed
            # workload (262,144 rays each, the same seeded draws) in every other arithmetic against the exact-f32 product frame --
            # |PSNR(., target) - PSNR(f32 frame, target)| for a uniform-random target (SURVEY 8(d)'s) AND for targets the f32 frame
            # approximates to 20 / 30 / 40 dB, on the whole frame and on scattered subsets of 3001 and 1024 rays; worst frame per cell
            try:
                from nerf import gate as GATE
                rows = {other: [] for other in others}
                for f in range(min(4, n_frames)):
                    frames = {}
                    for prec in [args.precision] + others:
                        nerf.set_mlp_precision(prec)
                        torch.manual_seed(4321 + f)
                        frames[prec] = step(f)[3]
                    for other in others:
                        r = GATE.gate_cells(frames["f32"], frames[other], seed=11 + f)
                        r["frame"] = f
                        r["max_abs_rgb_diff"] = float((frames[other].double() - frames["f32"].double()).abs().max())
                        rows[other].append(r)
                    del frames
                for other in others:
                    w = GATE.worst_of(rows[other])
                    line[key_of[other]]["gate_vs_exact_f32"] = {
                        "rays": H * W, "frames_checked": w["frames"], "min_self_psnr_db": w["min_self_psnr_db"], "max_self_psnr_db": w["max_self_psnr_db"],
                        "worst_abs_dpsnr_db": w["cells"], "gate_db": GATE.GATE_DB,
                        "passes": {t: {m: bool(v <= GATE.GATE_DB) for m, v in row.items()} for t, row in w["cells"].items()},
                        "what": "worst |PSNR(arithmetic, target) - PSNR(exact f32 frame, target)| over the frames, per target (uniform random; "
                                "20 / 30 / 40 dB = clamp(f32 frame + sigma randn)) and ray count (whole frame; worst of 8 scattered subsets of "
                                "3001 / 1024 rays)", "per_frame": rows[other]}
            except Exception as e:                                # an extra must never cost the headline
                line["whole_frame_parity_error"] = repr(e)
            nerf.set_mlp_precision(args.precision)
            torch.manual_seed(1234 + rank)

    # ---- configs[2] (N>1: configs[4]): training iterations in both arithmetics --------------------------------------------
    if not args.no_extras:
        train = {}
        keep_steps, keep_warm, keep_prec = args.steps, args.warmup, args.precision
        for prec in ("f32", "f16x3", "bf16x3"):
            args.steps, args.warmup, args.precision = args.train_steps, 5, prec
            nerf.set_mlp_precision(prec)
            mc_t, mf_t = synth_params(0, dev, "paper"), synth_params(1, dev, "paper")
            r = bench_train(args, nerf, mc_t, mf_t, dev, rank, world, dist, emit=False)
            if r is not None:
                train[prec] = {"value": r["value"], "unit": r["unit"], "ms_per_iter": r["ms_per_step"], "iters": r["steps"],
                               "warmup": r["warmup"], "roofline": r["roofline"], "allreduce": r["allreduce"]}
                train["allreduce"] = r["allreduce"]                       # (the same collective in every arithmetic: fp32 gradients)
        args.steps, args.warmup, args.precision = keep_steps, keep_warm, keep_prec
        nerf.set_mlp_precision(args.precision)
        if rank == 0:
            train["workload"] = ("configs[2]: paper-model training iteration, 2048 rays from a 512x512 frame, 64+64 samples, noise 0.1, "
                                 "latent table 1000x32, fwd + bwd + fused Adam" + ("" if world == 1 else
                                 f"; configs[4]: data parallel over {world} GPUs, one frame per rank, flat gradient all-reduce (RCCL)"))
            line["train"] = train
            if world == 1:
                pmc_train_traffic(train)
                train["clock_detail"] = pmc_train_clocks(train)

    if rank == 0:
        # ---- roofline of the dominant kernel: fused MLP forward, fine pass of one ray chunk ----------------
        S = N_COARSE + N_FINE
        pk = model_f.hip_weights().get()
        cond = ops.paper_condition(pk, conds[0][0], conds[0][1], NEAR, FAR)
        ro, rd = nerf.get_ray_bundle(H, W, INTRINSICS, poses[0])
        ro, rd = ro.view(-1, 3)[:CHUNK].contiguous(), rd.view(-1, 3)[:CHUNK].contiguous()
        z = torch.sort(torch.rand((CHUNK, S), device=dev) * (FAR - NEAR) + NEAR, dim=-1)[0].contiguous()
        pk_b = model_f.hip_weights().get_bf16()

        def timed(fn, n_launch=8):
            for _ in range(2):
                fn()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_launch + 1)]   # recorded on the launch stream
            ev[0].record()
            for k in range(n_launch):
                fn()
                ev[k + 1].record()
            torch.cuda.synchronize()
            return sum(ev[k].elapsed_time(ev[k + 1]) for k in range(n_launch)) / n_launch

        flops = float(CHUNK) * S * FLOP_PER_POINT
        algo_bytes = CHUNK * S * (4 + 16) + 2 * CHUNK * 12 + 4 * ops.H.lib().nf_paper_packed_floats()   # z read + raw written, rays, weights once
        pk_h = model_f.hip_weights().get_f16()
        ms = {"f32": timed(lambda: ops.paper_mlp_fwd(pk, cond, ro, rd, z)),
              "bf16x3": timed(lambda: ops.paper_mlp_fwd_bf16(pk_b, cond, ro, rd, z)),
              "f16x3": timed(lambda: ops.paper_mlp_fwd_f16(pk_h, cond, ro, rd, z)),
              "f16x2": timed(lambda: ops.paper_mlp_fwd_f16x2(pk_h, cond, ro, rd, z))}
        exe_f32 = float(CHUNK) * S * EXEC_FLOP_PER_POINT_F32 / (ms["f32"] * 1e-3) / 1e12
        objs = {"f32": {"bound": "mfma", "kernel": "k_paper_mlp_fwd<2> (65536 rays x 192 samples per launch)",
                        "achieved": flops / (ms["f32"] * 1e-3) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": exe_f32 / PEAK_F32_MFMA_TFLOPS,
                        "frac_algorithmic": flops / (ms["f32"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                        "avg_launch_ms": ms["f32"],
                        "algorithmic_flops_per_launch": flops, "algorithmic_hbm_bytes_per_launch": algo_bytes, "traffic": None,
                        "executed_tflops": exe_f32, "frac_executed": exe_f32 / PEAK_F32_MFMA_TFLOPS,
                        "frac_at_sustained_clock": None, "sustained_clock_mhz": None,
                        "note": "achieved counts the ALGORITHMIC FLOPs of the reference MLP (1,100,032 per point); the kernel folds the per-frame "
                                "constant input columns (expression, latent, PE(near), PE(far)) into bias vectors and issues 999,936 MFMA FLOPs per "
                                "point.  frac = executed_tflops / peak: the physical busy fraction of the fp32 matrix pipe at the NOMINAL 2.4 GHz clock "
                                "(<= 1); frac_algorithmic = achieved / peak may exceed 1 (the folded 9 % cost no matrix cycles); "
                                "frac_at_sustained_clock = executed FLOPs / (busy cycles of ONE PMC dispatch x 256 CUs x 256 FLOP/clk): cycles and "
                                "FLOPs of the same launch (GRBM_GUI_ACTIVE, a PMC pass of this run)",
                        "note_short": "frac = issued MFMA FLOPs (999,936/pt) / peak at 2.4 GHz; frac_algorithmic counts 1,100,032/pt"}}
        for prec, kname, peak_name in (("bf16x3", "k_paper_mlp_fwd_bf16", "bf16"), ("f16x3", "k_paper_mlp_fwd_f16", "fp16"),
                                       ("f16x2", "k_paper_mlp_fwd_f16x2", "fp16")):
            ach = flops / (ms[prec] * 1e-3) / 1e12
            exe = float(CHUNK) * S * BF16X3_EXEC_FLOP_PER_POINT * (SPLIT_MFMAS_PER_TILE["x2"] / float(SPLIT_MFMAS_PER_TILE["x3"]) if prec == "f16x2" else 1.0) / (ms[prec] * 1e-3) / 1e12
            objs[prec] = {"bound": "mfma", "kernel": f"{kname} (65536 rays x 192 samples per launch)",
                          "achieved": ach, "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_MFMA_TFLOPS,
                          "avg_launch_ms": ms[prec], "algorithmic_flops_per_launch": flops, "algorithmic_hbm_bytes_per_launch": algo_bytes,
                          "traffic": None, "executed_tflops": exe, "frac_executed": exe / PEAK_BF16_MFMA_TFLOPS,
                          "note": f"achieved counts ALGORITHMIC f32 FLOPs (1,100,032/point); each costs {2 if prec == 'f16x2' else 3} {peak_name} MFMA FLOPs in the split "
                                  f"scheme, so frac (against the dense {peak_name} peak) cannot exceed 1/{2 if prec == 'f16x2' else 3}; executed_* counts the issued MFMA FLOPs"
                                  + ("; against the fp32-MFMA peak (157.3 TFLOP/s) the same algorithmic rate is "
                                     f"{ach / PEAK_F32_MFMA_TFLOPS:.2f}x -- fp32-class results faster than the fp32 matrix pipe can issue them"
                                     if prec == "f16x3" else "")}
        if world == 1 and not args.no_extras:
            for prec in ("f32", "f16x3", "f16x2", "bf16x3"):
                objs[prec]["traffic"], objs[prec]["traffic_detail"] = pmc_traffic(prec)
            for prec in ("f16x3", "f16x2", "bf16x3"):
                objs[prec]["sustained_clock_mhz"], objs[prec]["sustained_clock_detail"] = pmc_sustained_clock(prec)
                if objs[prec]["sustained_clock_mhz"]:
                    objs[prec]["frac_executed_at_sustained_clock"] = objs[prec]["frac_executed"] * 2400.0 / objs[prec]["sustained_clock_mhz"]
            mhz, clk_detail = pmc_sustained_clock("f32")
            objs["f32"]["sustained_clock_mhz"], objs["f32"]["sustained_clock_detail"] = mhz, clk_detail
            if mhz:
                # ONE pass, one dispatch: executed FLOPs of the launch / (busy cycles of that dispatch x 256 CUs x 256 FLOP per clock) --
                # no HIP-event time of another run enters (the round-4 figure mixed two passes)
                cus = line["config"]["device"]["compute_units"]
                cyc = clk_detail["busy_cycles_raw_median"] / clk_detail["xcd_sum_divisor"]
                objs["f32"]["peak_at_sustained_clock_tflops"] = cus * 256 * mhz * 1e6 / 1e12
                objs["f32"]["frac_at_sustained_clock"] = float(CHUNK) * S * EXEC_FLOP_PER_POINT_F32 / (cyc * cus * 256)
        line["roofline"] = objs[args.precision]
        for other in others:
            line.setdefault(key_of[other], {})["roofline"] = objs[other]
        if world == 1 and not args.no_extras:
            try:
                line["tiny"], tiny_inputs = bench_tiny(dev)
                if not args.no_cpu_baseline:
                    line["tiny"]["cpu_baseline"] = cpu_baseline_tiny(*tiny_inputs)
            except Exception as e:                                # an extra must never cost the headline
                line["tiny"] = {"error": repr(e)}
            try:                                                  # configs[0] read literally: "4-layer MLP" (FlexibleNeRFModel, M:351-422)
                line["tiny4"], tiny_inputs = bench_tiny(dev, flex_layers=4)
                if not args.no_cpu_baseline:
                    line["tiny4"]["cpu_baseline"] = cpu_baseline_tiny(*tiny_inputs, flex_layers=4)
            except Exception as e:
                line["tiny4"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                torch.cuda.empty_cache()
                port = eager_rocm_baseline(dev)                   # the oracle's ops on the GPU (kind "port"), kept beside the real thing
            except Exception as e:                                # an extra must never cost the headline
                port = {"error": repr(e)}
            try:
                torch.cuda.empty_cache()
                line["eager_rocm"] = eager_rocm_reference(dev)    # raises where neither /root/reference nor oracle/_ref is present
                line["eager_rocm"]["port"] = port
            except Exception as e:
                line["eager_rocm"] = {**port, "reference_error": repr(e)}
            if line["eager_rocm"].get("value"):
                line["eager_rocm"]["product_over_eager"] = line["value"] / line["eager_rocm"]["value"]
            torch.cuda.empty_cache()
            line["cpu_baseline"] = cpu_baseline(args.cpu_rays)
            par = line["cpu_baseline"].get("parity_on_sample", {})
            for other in others:
                if isinstance(par, dict) and other in par and key_of[other] in line:
                    line[key_of[other]]["parity_on_sample"] = par[other]
        if world == 1 and not args.no_extras:
            try:
                torch.cuda.empty_cache()
                line["launcher"] = launcher_eval_leg(dev, model_c, model_f, args.launcher_frames)
            except Exception as e:                                # an extra must never cost the headline
                line["launcher"] = {"error": repr(e)}
            line["config"]["device"]["pattern_store"] = pattern_store_probe()
            try:
                line["config"]["device"]["power"] = power_probe(dev, (("f32", lambda: ops.paper_mlp_fwd(pk, cond, ro, rd, z)),
                                                                      ("f16x3", lambda: ops.paper_mlp_fwd_f16(pk_h, cond, ro, rd, z)),
                                                                      ("f16x2", lambda: ops.paper_mlp_fwd_f16x2(pk_h, cond, ro, rd, z)),
                                                                      ("bf16x3", lambda: ops.paper_mlp_fwd_bf16(pk_b, cond, ro, rd, z)),
                                                                      # the training forward with saves at 262144 points, split-bf16 and exact f32:
                                                                      # the kernels the driver's boxes of rounds 3 and 4 ran 2.9x / 1.17x slower
                                                                      ("train_fwd_bf16x3", lambda: ops.paper_mlp_fwd_train(pk, cond, ro[:2048], rd[:2048], z[:2048, :128].contiguous(), packed_b=pk_b)),
                                                                      ("train_fwd_f32", lambda: ops.paper_mlp_fwd_train(pk, cond, ro[:2048], rd[:2048], z[:2048, :128].contiguous()))))
            except Exception as e:
                line["config"]["device"]["power"] = {"error": repr(e)}
        line["ranks_seen"] = int(dist.get_world_size()) if dist is not None else 1
        line["per_rank_ms_per_step"] = per_rank_ms
        line["summary"] = summary_of(line)
        emit(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


COMPACT_LIMIT = 6144            # bytes: the driver's record keeps ~8 KB of stdout tail; the final line must fit with room to spare
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_algorithmic", "avg_launch_ms", "traffic",
                 "algorithmic_hbm_bytes_per_launch", "sustained_clock_mhz", "frac_at_sustained_clock")
CPU_BASELINE_KEYS = ("value", "unit", "kind", "cores", "host_cores", "sample")


def _sig(x, digits=6):
    """Floats to `digits` significant digits (the compact line is a record, not a checkpoint); containers recursively."""
    if isinstance(x, float):
        return float(f"{x:.{digits}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(line):
    """The ONE final stdout line (<= COMPACT_LIMIT bytes): the contract's keys, `roofline` and `cpu_baseline` as flat objects,
    `summary` (flat scalars).  Everything else (per-kernel objects, *_detail, parity_on_sample, tiny, train ...) is in the detail
    line printed before it and in bench_detail.json."""
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: line.get(k) for k in head}
    out["config"] = {"workload": (line.get("config") or {}).get("workload")}
    out["ranks_seen"] = line.get("ranks_seen")
    rf = line.get("roofline")
    if isinstance(rf, dict):
        out["roofline"] = {k: rf.get(k) for k in ROOFLINE_KEYS if k in rf}
        if isinstance(rf.get("note_short", rf.get("note")), str):
            out["roofline"]["note"] = rf.get("note_short", rf["note"])[:120]
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict):
        out["cpu_baseline"] = {k: cb.get(k) for k in CPU_BASELINE_KEYS}
        if isinstance(out["cpu_baseline"].get("sample"), str):
            out["cpu_baseline"]["sample"] = out["cpu_baseline"]["sample"][:200]
    out["detail"] = "bench_detail.json (and the stdout line before this one)"
    summary = _sig({k: v for k, v in (line.get("summary") or {}).items() if v is not None})
    out = _sig(out)
    if len(json.dumps({**out, "summary": summary})) >= COMPACT_LIMIT:      # never lose the record to an overgrown summary: shed it from the back
        out["summary_truncated"] = True
        keys = list(summary)
        while keys and len(json.dumps({**out, "summary": summary})) >= COMPACT_LIMIT:
            summary.pop(keys.pop())
    out["summary"] = summary                                                # LAST key
    return out


def emit(line):
    """Detail first (one stdout line + bench_detail.json), the compact record LAST."""
    detail = json.dumps({"bench_detail": line})
    for d in (os.path.join(ROOT, "gpurun_out"), ROOT):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_detail.json"), "w") as f:
                    f.write(detail + "\n")
            except OSError:
                pass
    print(detail, flush=True)
    print(json.dumps(compact_line(line)), flush=True)


def summary_of(line):
    """Flat scalars of the compact line: everything a reader needs to check the claims of DESIGN.md against the driver's own run.
    Key names are short on purpose (the whole record must stay below COMPACT_LIMIT); tags: f16 = f16x3, x2 = f16x2, bf = bf16x3."""
    def g(*path, default=None):
        o = line
        for k in path:
            if not isinstance(o, dict) or k not in o:
                return default
            o = o[k]
        return o
    per = line.get("per_rank_ms_per_step") or []
    s = {"value_rays_s": line.get("value"), "ms_per_step": line.get("ms_per_step"), "n_gpus": line.get("n_gpus"),
         "ranks_seen": line.get("ranks_seen"), "rank_ms_min": min(per) if per else None, "rank_ms_max": max(per) if per else None,
         "roofline_frac": g("roofline", "frac"), "roofline_avg_launch_ms": g("roofline", "avg_launch_ms")}
    for tag, key in (("f16", "split_f16"), ("x2", "split_f16x2"), ("bf", "split_bf16")):
        s[f"{tag}_rays_s"] = g(key, "value")
        s[f"{tag}_launch_ms"] = g(key, "roofline", "avg_launch_ms")
        s[f"{tag}_clock_mhz"] = g(key, "roofline", "sustained_clock_mhz")
        s[f"{tag}_frac_executed"] = g(key, "roofline", "frac_executed")
        # north_star's gate per arithmetic on the first frames of the workload (worst frame): SURVEY's random target, a target the
        # f32 frame approximates to 30 dB (whole frame; worst 1024-ray subset), lowest self-PSNR against the f32 frame
        cells = g(key, "gate_vs_exact_f32", "worst_abs_dpsnr_db") or {}
        s[f"{tag}_gate_random_db"] = (cells.get("random") or {}).get("whole")
        s[f"{tag}_gate_30db_worst_db"] = (cells.get("30dB") or {}).get("whole")
        s[f"{tag}_gate_30db_1024rays_db"] = (cells.get("30dB") or {}).get("1024")
        s[f"{tag}_gate_40db_worst_db"] = (cells.get("40dB") or {}).get("whole")
        s[f"{tag}_self_psnr_min_db"] = g(key, "gate_vs_exact_f32", "min_self_psnr_db")
    s["gate_frames"] = g("split_f16", "gate_vs_exact_f32", "frames_checked")
    for prec, tag in (("f32", "f32"), ("f16x3", "f16"), ("bf16x3", "bf")):
        s[f"train_ms_{tag}"] = g("train", prec, "ms_per_iter")
        ks = g("train", prec, "roofline", "kernels", default=[]) or []
        for kt, k in zip(("fwd", "chain", "dw"), ks):
            s[f"tr_{tag}_{kt}_ms"] = k.get("avg_launch_ms")
            s[f"tr_{tag}_{kt}_frac"] = k.get("frac")                       # f32: executed / fp32-MFMA peak; split: algorithmic bytes / 8 TB/s
            if prec != "f32":
                s[f"tr_{tag}_{kt}_frac_mfma"] = k.get("frac_executed_mfma")   # issued 16-bit MFMA FLOPs / 2.5 PFLOP/s
            s[f"tr_{tag}_{kt}_mhz"] = k.get("sustained_clock_mhz")
            s[f"tr_{tag}_{kt}_ms_2400_est"] = k.get("ms_at_nominal_clock")    # ESTIMATE: busy cycles of ONE PMC dispatch / 2.4 GHz
    ar = g("train", "allreduce") or {}
    ps = g("config", "device", "pattern_store") or {}
    s.update({"train_allreduce_us": ar.get("allreduce_us"), "train_bytes_allreduced": ar.get("bytes_allreduced"),
              "train_ranks_seen": ar.get("ranks_seen"),
              "hbm_fill_gbs": g("config", "device", "hbm_fill_gbs"), "pattern_store_gbs": ps.get("stream_nt_gbs"),
              "engine_clock_mhz": g("config", "device", "engine_clock_mhz"),
              "eager_rocm_rays_s": g("eager_rocm", "value"), "eager_rocm_kind": g("eager_rocm", "kind"),
              "eager_rocm_min_rays_s": g("eager_rocm", "value_min"), "eager_rocm_max_rays_s": g("eager_rocm", "value_max"),
              "eager_rocm_frames": g("eager_rocm", "frames"),
              "product_over_eager": g("eager_rocm", "product_over_eager"),
              "cpu_baseline_rays_s": g("cpu_baseline", "value"), "cpu_baseline_kind": g("cpu_baseline", "kind"),
              "cpu_threads": g("cpu_baseline", "cores"),
              "tiny_rays_s": g("tiny", "value"), "tiny_cpu_rays_s": g("tiny", "cpu_baseline", "value"),
              "tiny_cpu_kind": g("tiny", "cpu_baseline", "kind"),
              "tiny4_rays_s": g("tiny4", "value"), "tiny4_cpu_rays_s": g("tiny4", "cpu_baseline", "value"),
              "launcher_eval_frames_s": g("launcher", "launcher_eval_frames_s"),
              "launcher_wall_over_gpu": g("launcher", "launcher_wall_over_gpu")})
    # north_star's ratio "x the reference on the same GPU at matched PSNR": quoted on f16x3, the fastest arithmetic that keeps the gate at
    # realistic targets (profiles/r06_gate_sensitivity.md); f16x2 / bf16x3 hold it against SURVEY's random target on whole frames only, so
    # their ratios are SPEED figures, not matched-PSNR ones
    if s.get("eager_rocm_rays_s"):
        for tag in ("f16", "x2", "bf"):
            if s.get(f"{tag}_rays_s"):
                s[{"f16": "matched_psnr_over_eager_f16x3", "x2": "speed_only_over_eager_f16x2", "bf": "speed_only_over_eager_bf16x3"}[tag]] = \
                    s[f"{tag}_rays_s"] / s["eager_rocm_rays_s"]
    pw = g("config", "device", "power") or {}
    st = pw.get("static") or {}
    s.update({"power_cap_w": st.get("power_cap_w"), "perf_level": st.get("perf_level")})
    for prec, tag in (("f32", "f32"), ("f16x3", "f16"), ("f16x2", "x2"), ("bf16x3", "bf"), ("train_fwd_bf16x3", "trfwd_bf"), ("train_fwd_f32", "trfwd_f32")):
        o = pw.get(prec) or {}                                        # how THIS box holds its power cap under each kernel
        s.update({f"w_{tag}": o.get("power_w"), f"mhz_{tag}": o.get("sclk_mhz_hwmon") or o.get("sclk_mhz_dpm")})
    return s


if __name__ == "__main__":
    main()
