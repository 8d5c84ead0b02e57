"""Data-parallel NeRFace trainer: the loop of the reference's train_transformed_rays.py (TR:243-573) with one frame
per rank per step and one flat RCCL gradient all-reduce.

    torchrun --standalone --nproc-per-node 8 4d-facial-avatars_amd/launch/train_sharded.py --config cfg.yml [--load-checkpoint ckpt]

Same CLI, YAML schema and checkpoint dictionary as the reference (`iter, model_coarse_state_dict, model_fine_state_dict,
optimizer_state_dict, loss, psnr, background, latent_codes`; two optimizer param groups).  Differences, all outside the
hot path: ray selection (importance map, p = 0.9 inside the bbox, TR:230-239) draws on the device (nerf.choose_rays: a HIP
radix select over exponential keys, torch's generator) instead of np.random.choice on the host; rank 0 writes checkpoints and
the reference's TensorBoard scalars / validation images (TR:415-424, 518-541; a JSON-lines file when the `tensorboard` package
is absent), accumulated on the device and flushed on the iterations that print; on resume the latent codes and
background are restored *into* the tensors the optimizer already owns (the reference re-wraps them and the optimizer
keeps stepping the stale ones).
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

from . import common as CM
from .common import D, nerf


def importance_maps(bboxs, H, W, p=0.9):
    """Per-pixel selection weights, indexed by the ROW-MAJOR pixel index row * W + col that nerf.choose_rays / nerf.get_ray_batch use.
    The reference draws an index k with probability probs.reshape(-1)[k] (TR:230-239: the (H, W) map, p inside the bbox rows
    b0:b1 x columns b2:b3) and then looks the pixel up in coords = meshgrid_xy(arange(H), arange(W)).reshape(-1, 2) (TR:302-330),
    whose k-th entry is (row = k % H, col = k // H) -- the map is applied TRANSPOSED.  A drop-in keeps that training
    distribution: weight(row, col) = probs.reshape(-1)[col * H + row]."""
    maps = []
    for b in bboxs:
        m = np.full((H, W), 1 - p, dtype=np.float64)
        m[int(b[0]):int(b[1]), int(b[2]):int(b[3])] = p
        flat = (m / m.sum()).reshape(-1)                                   # the reference's probs[img_idx]
        maps.append(np.ascontiguousarray(flat.reshape(W, H).T).reshape(-1))   # [row * W + col] = flat[col * H + row]
    return maps


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, required=True, help="Path to (.yml) config file.")
    ap.add_argument("--load-checkpoint", type=str, default="", help="Path to load saved checkpoint from.")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=os.environ.get("NERFACE_DIST_BACKEND", "nccl"),
                    help="torch.distributed backend: nccl = RCCL over xGMI (default); gloo = several ranks on one GPU (tests)")
    ap.add_argument("--precision", choices=["f32", "f16x3", "bf16x3"], default=os.environ.get("NERFACE_MLP_PRECISION", "f32"),
                    help="arithmetic of the three training GEMM kernels: f32 (default) = the reference's; f16x3 = split-fp16, gradients "
                         "at least as accurate as the f32 kernels' at 2.4x the speed; bf16x3 = split-bf16, fastest (1e-4 per tensor)")
    args = ap.parse_args(argv)
    rank, world, dev = CM.init_distributed(args.backend)
    nerf.set_mlp_precision(args.precision)
    cfg = CM.load_config(args.config)
    images, poses, render_poses, hwf, i_split, expressions, _, bboxs = nerf.load_flame_data(
        cfg.dataset.basedir, half_res=cfg.dataset.half_res, testskip=cfg.dataset.testskip)
    i_train, i_val, i_test = i_split
    H, W, intrinsics = int(hwf[0]), int(hwf[1]), hwf[2]
    # frames stay on the host in pageable memory (the reference keeps the whole sequence in one CPU tensor, TR:61-72); the one
    # frame a step needs crosses PCIe asynchronously through a pinned double buffer, overlapped with the previous step's kernels
    stager = CM.FrameStager(images, poses[:, :3, :4].contiguous(), expressions, dev)
    seed = cfg.experiment.randomseed
    np.random.seed(D.rank_seed(seed))
    torch.manual_seed(seed)                                   # identical model init everywhere (also broadcast below)
    enc_xyz, enc_dir = CM.build_encoders(cfg)
    model_c, model_f = CM.build_models(cfg, dev)
    background = CM.load_background(cfg.dataset.basedir, H, W, dev)
    latent_codes = torch.zeros(len(i_train), 32, device=dev, requires_grad=True)
    trainable = list(model_c.parameters()) + (list(model_f.parameters()) if model_f is not None else []) + [latent_codes]
    groups = [{"params": trainable}]
    if background is not None:
        groups.append({"params": background, "lr": cfg.optimizer.lr})          # inert 2nd group, kept for checkpoint compatibility
    # TR:193-199 `getattr(torch.optim, cfg.optimizer.type)`; nerf.optim holds one-launch forms of the same update rule (Adam) with
    # torch's state layout, so checkpoints stay interchangeable with the reference's
    optimizer = (getattr(nerf.optim, cfg.optimizer.type, None) or getattr(torch.optim, cfg.optimizer.type))(groups, lr=cfg.optimizer.lr)
    start_iter = 0
    if args.load_checkpoint and os.path.exists(args.load_checkpoint):
        ck = torch.load(args.load_checkpoint, map_location=dev)
        model_c.load_state_dict(ck["model_coarse_state_dict"])
        if ck.get("model_fine_state_dict") and model_f is not None:
            model_f.load_state_dict(ck["model_fine_state_dict"])
        if ck.get("latent_codes") is not None:
            with torch.no_grad():
                latent_codes.copy_(ck["latent_codes"])
        if ck.get("background") is not None and background is not None:
            background.copy_(ck["background"])
        optimizer.load_state_dict(ck["optimizer_state_dict"])
        start_iter = int(ck["iter"])
    D.broadcast_parameters(trainable)
    # from here on every rank draws its OWN rays and noise: the shared seed above served the identical initialisation only
    torch.manual_seed(D.rank_seed(seed))
    reducer = D.GradientAllReducer(trainable)
    maps = [torch.from_numpy(m).to(device=dev, dtype=torch.float32) for m in importance_maps(bboxs[i_train].numpy(), H, W)]
    logdir = os.path.join(cfg.experiment.logdir, cfg.experiment.id)
    if rank == 0:
        os.makedirs(logdir, exist_ok=True)
        with open(os.path.join(logdir, "config.yml"), "w") as f:
            f.write(cfg.dump())
    model_c.train()
    if model_f is not None:
        model_f.train()
    log = CM.ScalarLog(logdir) if rank == 0 else None
    if rank == 0:
        print(f"[LOG] scalars -> {log.kind} ({logdir})")
    f16 = args.precision == "f16x3"
    if f16:
        # split-fp16 range guard, armed for training: the exact-f32 probe of the hidden activations runs on a model's first step
        # and every print_every-th after it, and the kernels' sticky non-finite flag is polled on the same iterations and at
        # every checkpoint (the only iterations that synchronise with the host anyway)
        nerf.ops.set_f16_train_probe_every(cfg.experiment.print_every)
    n_rays = cfg.nerf.train.num_random_rays
    t0 = time.time()
    first_draw = None
    for i in range(start_iter, cfg.experiment.train_iters):
        k = int(np.random.randint(len(i_train)))              # one frame per rank per step (TR:289)
        img_idx = int(i_train[k])
        target_img, pose, expr = stager.fetch(img_idx)
        latent = latent_codes[k]
        sel = nerf.choose_rays(maps[k], n_rays)               # TR:320-322 on the device: n distinct pixels, p = the importance map
        if first_draw is None:
            first_draw = (img_idx, sel[:16].clone())
        # rays, target pixels and background prior of the selected pixels only, one kernel (TR:302 builds the full 512 x 512
        # bundle every iteration and gathers four times, TR:325-330)
        ro, rd, target, bg = nerf.get_ray_batch(H, W, intrinsics, pose, sel, target_img, background)
        rgb_c, _, _, rgb_f, _, _, _ = nerf.run_one_iter_of_nerf(
            H, W, intrinsics, model_c, model_f, ro, rd, cfg, mode="train", encode_position_fn=enc_xyz,
            encode_direction_fn=enc_dir, expressions=expr, background_prior=bg, latent_code=latent)
        # TR:355-387 (coarse + fine mse, 10 x 0.0005 x ||latent||) and -- in loss.backward() -- the gradients of those nodes: two launches
        # (nerf.training_loss) instead of ~20; parts = [loss, coarse, fine, code loss, coarse + fine, its PSNR, ||latent||] stay on the device
        # and are read back (a host sync) only on the iterations that log or save
        loss, parts = nerf.training_loss(rgb_c[..., :3], rgb_f[..., :3] if rgb_f is not None else None, target[..., :3], latent)
        coarse_loss, fine_loss, code_loss, mse = parts[1], (parts[2] if rgb_f is not None else None), parts[3], parts[4]
        loss.backward()
        reducer.reduce()
        optimizer.step()
        optimizer.zero_grad()
        lr_new = cfg.optimizer.lr * (cfg.scheduler.lr_decay_factor ** (i / (cfg.scheduler.lr_decay * 1000)))
        for g in optimizer.param_groups:
            g["lr"] = lr_new
        logs_now = i % cfg.experiment.print_every == 0 or i == cfg.experiment.train_iters - 1
        saves_now = i % cfg.experiment.save_every == 0 or i == cfg.experiment.train_iters - 1
        if f16 and (logs_now or saves_now):
            # every rank raises TOGETHER (flag MAX-reduced) if any rank's step since the last poll overflowed fp16 -- before the
            # checkpoint below is written
            nerf.ops.check_f16_range(model_c, model_f, sync_ranks=True)
        if rank == 0:
            # TR:415-424, device-side: the scalars of every iteration, read back together when the iteration prints
            log.add_scalar("train/code_loss", code_loss, i)
            log.add_scalar("train/coarse_loss", coarse_loss, i)
            if fine_loss is not None:
                log.add_scalar("train/fine_loss", fine_loss, i)
            log.add_scalar("train/psnr", parts[5], i)
        if rank == 0 and logs_now:
            log.flush()
            print(f"[TRAIN] Iter: {i} Loss: {loss.item():.6f} PSNR: {nerf.mse2psnr(mse.item()):.4f} "
                  f"({(time.time() - t0):.1f} s, {world} GPU)")
        if i % cfg.experiment.validate_every == 0:
            # TR:427-505: every validate_every iterations the first two validation frames are rendered whole (validation chunking and
            # sample counts) with a ZERO latent code and -- as the reference does -- the expression of the training frame just used;
            # loss = sum over the frames of 2 x fine mse (coarse mse without a fine model), divided by len(i_val) (TR:498-503).
            # The two frames go to different ranks; one scalar is all-reduced.
            model_c.eval()
            if model_f is not None:
                model_f.eval()
            with torch.no_grad():
                val_sum = torch.zeros(1, device=dev)
                val_parts = torch.zeros(2, device=dev)                     # [sum of coarse mse, sum of fine mse] (TR:518-541)
                val_img = None
                for j, v_idx in enumerate(i_val[:2]):
                    if j % world != rank:
                        continue
                    v_img, v_pose, _ = stager.fetch(int(v_idx))
                    v_ro, v_rd = nerf.get_ray_bundle(H, W, intrinsics, v_pose)
                    v_out = nerf.run_one_iter_of_nerf(
                        H, W, intrinsics, model_c, model_f, v_ro, v_rd, cfg, mode="validation", encode_position_fn=enc_xyz,
                        encode_direction_fn=enc_dir, expressions=expr,
                        background_prior=background.view(-1, 3) if background is not None else None,
                        latent_code=torch.zeros(32, device=dev))
                    v_coarse = nerf.img2mse(v_out[0][..., :3], v_img[..., :3])
                    v_loss = v_coarse
                    val_parts[0] += v_coarse
                    if v_out[3] is not None:
                        v_fine = nerf.img2mse(v_out[3][..., :3], v_img[..., :3])
                        v_loss = 2.0 * v_fine
                        val_parts[1] += v_fine
                    val_sum += v_loss
                    if rank == 0 and val_img is None:
                        val_img = (v_out[0], v_out[3], v_img)
                if torch.distributed.is_initialized() and not D._skip_collectives(world):
                    torch.distributed.all_reduce(val_sum)
                    torch.distributed.all_reduce(val_parts)
                val_loss = float(val_sum) / max(len(i_val), 1)
                if rank == 0:
                    log.add_scalar("validation/loss", val_loss, i)
                    log.add_scalar("validation/coarse_loss", val_parts[0] / max(len(i_val), 1), i)
                    log.add_scalar("validation/psnr", nerf.mse2psnr(val_loss), i)
                    if model_f is not None:
                        log.add_scalar("validation/fine_loss", val_parts[1] / max(len(i_val), 1), i)
                    if val_img is not None:                                # TR:528-541: images of the first validation frame
                        log.add_image("validation/rgb_coarse", val_img[0][..., :3].permute(2, 0, 1), i)
                        if val_img[1] is not None:
                            log.add_image("validation/rgb_fine", val_img[1][..., :3].permute(2, 0, 1), i)
                        log.add_image("validation/img_target", val_img[2][..., :3].permute(2, 0, 1), i)
                        if background is not None:
                            log.add_image("validation/background", background[..., :3].permute(2, 0, 1), i)
                    log.flush()
            if rank == 0:
                print(f"[VAL] Iter: {i} Validation loss: {val_loss:.6f} Validation PSNR: {nerf.mse2psnr(val_loss):.4f} "
                      f"({(time.time() - t0):.1f} s)")
            model_c.train()
            if model_f is not None:
                model_f.train()
        if rank == 0 and saves_now:
            psnr = nerf.mse2psnr(mse.item())
            torch.save({"iter": i, "model_coarse_state_dict": model_c.state_dict(),
                        "model_fine_state_dict": None if model_f is None else model_f.state_dict(),
                        "optimizer_state_dict": optimizer.state_dict(), "loss": loss, "psnr": psnr,
                        "background": None if background is None else background.data, "latent_codes": latent_codes.data},
                       os.path.join(logdir, "checkpoint" + str(i).zfill(5) + ".ckpt"))
    if rank == 0:
        log.close()
    if torch.distributed.is_initialized():
        if first_draw is None:                                 # resumed at or past train_iters: the loop never ran
            first_draw = (-1, torch.full((16,), -1, dtype=torch.int64, device=dev))
        # replica consistency: after identical Adam steps on averaged gradients every rank must hold the same parameters, and
        # the ranks must have drawn different frames / rays (one checksum + the first draw per rank, gathered once at the end)
        import json
        chk = torch.stack([p.detach().double().sum() for p in trainable]).sum().reshape(1)
        draw = torch.cat((torch.tensor([float(first_draw[0])], device=dev), first_draw[1].reshape(-1).float())).double()
        both = torch.cat((chk, draw))
        gathered = [torch.zeros_like(both) for _ in range(world)]
        torch.distributed.all_gather(gathered, both)
        if rank == 0:
            sums = [float(g[0]) for g in gathered]
            draws = [[int(v) for v in g[1:].tolist()] for g in gathered]
            report = {"world": world, "iters": int(cfg.experiment.train_iters - start_iter), "parameter_checksums": sums,
                      "identical_parameters": all(v == sums[0] for v in sums), "first_draw_per_rank": draws,
                      "distinct_draws": len({tuple(d) for d in draws}) == world}
            with open(os.path.join(logdir, "dp_consistency.json"), "w") as f:
                json.dump(report, f)
            print(f"[DP] {world} ranks: identical parameters = {report['identical_parameters']}, distinct ray draws = {report['distinct_draws']}")
        torch.distributed.barrier()
    return logdir


if __name__ == "__main__":
    main()
