"""north_star's gate on a TRAINED model against its own training targets -- the case the reference computes PSNR for
(train_transformed_rays.py:355-392, 438-470): a "teacher" pair of paper models (seeded, density head x`head`) renders N frames of the synthetic
sequence at 512 x 512 with the exact-f32 kernels; a "student" pair (fresh default init, zero latent table) is trained on those frames with the
trainer's loop body (importance-free uniform ray draws, 2048 rays / iteration, 64 + 64 samples, noise 0.1, Adam 5e-4, exact f32); then every
training frame is rendered with the student in all four arithmetics (shipped validation settings: 64 + 128, perturb on, same seeded draws) and
|PSNR(arithmetic, teacher image) - PSNR(f32, teacher image)| is measured beside PSNR(f32, teacher image) and the self-PSNR.  The residual of a
trained model is STRUCTURED (not the white noise of nerf.gate.target_near), so this is the check that the synthetic-target table of
profiles/r06_gate_sensitivity.md carries over.  argv: iterations (default 4000), frames (default 6), head (default 40), output JSON path."""
import json, math, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
from nerf import gate as G

dev = torch.device("cuda:0")
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
head = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
out_path = sys.argv[4] if len(sys.argv) > 4 else os.path.join(R, "gpurun_out", "trained_gate.json")
H = W = 512


def model(seed, boost):
    torch.manual_seed(seed)
    m = nerf.models.ConditionalBlendshapePaperNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=False,
                                                        use_viewdirs=True, num_layers=4, hidden_size=256, include_expression=True)
    if boost:
        with torch.no_grad():
            m.fc_alpha.weight.mul_(boost); m.fc_alpha.bias.fill_(0.5 if boost <= 100 else 5.0); m.fc_rgb.weight.mul_(10.0)
    return m.to(dev)


ex = nerf.get_embedding_function(num_encoding_functions=10, include_input=True, log_sampling=True)
ed = nerf.get_embedding_function(num_encoding_functions=4, include_input=False, log_sampling=True)
bg_img = torch.rand((H, W, 3), generator=torch.Generator().manual_seed(7)).to(dev)
bg = bg_img.view(-1, 3)
poses = [bench.frame_pose(7 * f).to(dev) for f in range(n_frames)]
conds = []
for f in range(n_frames):
    g = torch.Generator().manual_seed(1000 + f)
    conds.append(((0.5 * torch.randn(76, generator=g)).to(dev), (0.1 * torch.randn(32, generator=g)).to(dev)))

# ---- the teacher's frames = the training targets ---------------------------------------------------------------------
nerf.set_mlp_precision("f32")
tc, tf = model(100, head).eval(), model(101, head).eval()
val = dict(num_coarse=64, num_fine=128, chunksize=65536, perturb=False, lindisp=False, radiance_field_noise_std=0.0, white_background=False)
trn = dict(num_coarse=64, num_fine=64, chunksize=2048, perturb=True, lindisp=False, radiance_field_noise_std=0.1, white_background=False)
opt = nerf.CfgNode(dict(nerf=dict(use_viewdirs=True, train=trn, validation=val), dataset=dict(no_ndc=True, near=bench.NEAR, far=bench.FAR)))
targets = []
with torch.no_grad():
    for f in range(n_frames):
        ro, rd = nerf.get_ray_bundle(H, W, bench.INTRINSICS, poses[f])
        targets.append(nerf.run_one_iter_of_nerf(H, W, bench.INTRINSICS, tc, tf, ro, rd, opt, mode="validation", encode_position_fn=ex, encode_direction_fn=ed,
                                                 expressions=conds[f][0], background_prior=bg, latent_code=conds[f][1])[3].contiguous())
del tc, tf

# ---- the student, trained on them with the trainer's iteration (TR:289-400) ----------------------------------------------
sc, sf = model(7, 0).train(), model(8, 0).train()
latents = torch.zeros(n_frames, 32, device=dev).requires_grad_(True)
params = list(sc.parameters()) + list(sf.parameters()) + [latents]
optim = nerf.optim.Adam(params, lr=5e-4)
torch.manual_seed(99)
t0 = time.time()
for it in range(n_iter):
    f = it % n_frames
    sel = torch.randint(0, H * W, (2048,), device=dev)
    ro, rd, tgt, bgp = nerf.get_ray_batch(H, W, bench.INTRINSICS, poses[f], sel, targets[f], bg_img)
    lat = latents[f]
    out = nerf.run_one_iter_of_nerf(H, W, bench.INTRINSICS, sc, sf, ro, rd, opt, mode="train", encode_position_fn=ex, encode_direction_fn=ed,
                                    expressions=conds[f][0], background_prior=bgp, latent_code=lat)
    loss, parts = nerf.training_loss(out[0], out[3], tgt, lat)
    loss.backward()
    optim.step()
    optim.zero_grad()
    if it % 500 == 0 or it == n_iter - 1:
        print(f"iteration {it}: loss {float(loss):.5f}, train PSNR (coarse + fine) {float(parts[5]):.2f} dB, {time.time() - t0:.0f} s", flush=True)
sc.eval(); sf.eval()

# ---- the gate of every arithmetic on the training frames, against the teacher's images ----------------------------------------
val["perturb"] = True                                              # the shipped validation block (CFG:149-167)
opt = nerf.CfgNode(dict(nerf=dict(use_viewdirs=True, train=trn, validation=val), dataset=dict(no_ndc=True, near=bench.NEAR, far=bench.FAR)))
rows = {p: [] for p in ("f16x3", "bf16x3", "f16x2")}
psnr_f32 = []
for f in range(n_frames):
    frames = {}
    ro, rd = nerf.get_ray_bundle(H, W, bench.INTRINSICS, poses[f])
    for prec in ("f32", "f16x3", "bf16x3", "f16x2"):
        nerf.set_mlp_precision(prec)
        torch.manual_seed(4321 + f)
        with torch.no_grad():
            frames[prec] = nerf.run_one_iter_of_nerf(H, W, bench.INTRINSICS, sc, sf, ro, rd, opt, mode="validation", encode_position_fn=ex, encode_direction_fn=ed,
                                                     expressions=conds[f][0], background_prior=bg, latent_code=latents[f].detach())[3]
    p32 = G.psnr(frames["f32"], targets[f])
    psnr_f32.append(p32)
    g = torch.Generator().manual_seed(55 + f)
    subsets = {m: [torch.randperm(H * W, generator=g)[:m].to(dev) for _ in range(8)] for m in G.SUBSET_RAYS}
    t = targets[f].reshape(-1, 3).double()
    se32 = ((frames["f32"].reshape(-1, 3).double() - t) ** 2).sum(1)
    for prec in rows:
        se = ((frames[prec].reshape(-1, 3).double() - t) ** 2).sum(1)
        d = lambda a, b: abs(10.0 * math.log10(float(a) / float(b)))
        r = {"frame": f, "psnr_f32_db": p32, "self_psnr_db": G.psnr(frames[prec], frames["f32"]), "whole": d(se.sum(), se32.sum())}
        for m, idxs in subsets.items():
            r[str(m)] = max(d(se[i].sum(), se32[i].sum()) for i in idxs)
        rows[prec].append(r)
    print(f"frame {f}: PSNR(student f32, teacher image) {p32:.2f} dB | " + " | ".join(
        f"{p} whole {rows[p][-1]['whole']:.1e} 3001 {rows[p][-1]['3001']:.1e} 1024 {rows[p][-1]['1024']:.1e} (self {rows[p][-1]['self_psnr_db']:.1f} dB)" for p in rows), flush=True)
nerf.set_mlp_precision("f32")
summary = {p: {k: max(r[k] for r in rows[p]) for k in ("whole", "3001", "1024")} | {"min_self_psnr_db": min(r["self_psnr_db"] for r in rows[p])} for p in rows}
print(f"\ntrained for {n_iter} iterations on {n_frames} frames (teacher head x{head:g}); PSNR(student f32, target) {min(psnr_f32):.2f} .. {max(psnr_f32):.2f} dB")
print("| arithmetic | self-PSNR min dB | whole frame | worst of 8 x 3001 rays | worst of 8 x 1024 rays |\n|---|---|---|---|---|")
for p, s in summary.items():
    b = lambda v: ("**%.1e**" if v > G.GATE_DB else "%.1e") % v
    print(f"| {p} | {s['min_self_psnr_db']:.1f} | {b(s['whole'])} | {b(s['3001'])} | {b(s['1024'])} |")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump({"iterations": n_iter, "frames": n_frames, "head": head, "psnr_f32_db": psnr_f32, "rows": rows, "summary": summary}, open(out_path, "w"))
