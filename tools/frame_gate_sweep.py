"""north_star's gate over MANY frames, realistic targets and small ray sets: for f16x3 / bf16x3 / f16x2 against the product's
exact-f32 frame (same seeded draws), frames 0..N-1 of bench.py's scene (x1000 density head) and of the same scene with SURVEY 8(d)'s x40 head,
every cell of nerf.gate.gate_cells -- targets {uniform random, 20 / 30 / 40 dB around the exact frame} x {whole 512 x 512 frame, 8 scattered
subsets of 3001 and of 1024 rays}: worst |PSNR(arithmetic, target) - PSNR(f32, target)| per cell and the self-PSNR range.
argv: number of frames per scene (default 8), output JSON path (default gpurun_out/gate_sweep.json).  Prints a markdown table."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
from nerf import gate as G

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(R, "gpurun_out", "gate_sweep.json")
PRECS = ("f16x3", "bf16x3", "f16x2")


def models(scene):
    mc, mf = bench.synth_params(0, dev), bench.synth_params(1, dev)
    if scene == "soft":                                        # fc_alpha x40, bias 0.5 (bench.synth_params applied x1000, bias 5)
        with torch.no_grad():
            for m in (mc, mf):
                m.fc_alpha.weight.mul_(40.0 / 1000.0)
                m.fc_alpha.bias.fill_(0.5)
    return mc, mf


opt = bench.options(nerf)
ex = nerf.get_embedding_function(num_encoding_functions=10, include_input=True, log_sampling=True)
ed = nerf.get_embedding_function(num_encoding_functions=4, include_input=False, log_sampling=True)
bg = torch.rand((512, 512, 3), generator=torch.Generator().manual_seed(7)).to(dev).view(-1, 3)
result = {"frames_per_scene": n, "scenes": {}}
for scene in ("bench", "soft"):
    mc, mf = models(scene)
    rows = {p: [] for p in PRECS}
    for f in range(n):
        g = torch.Generator().manual_seed(1000 + f)
        expr, lat = (0.5 * torch.randn(76, generator=g)).to(dev), (0.1 * torch.randn(32, generator=g)).to(dev)
        ro, rd = nerf.get_ray_bundle(512, 512, bench.INTRINSICS, bench.frame_pose(f).to(dev))
        frames = {}
        for prec in ("f32",) + PRECS:
            nerf.set_mlp_precision(prec)
            torch.manual_seed(4321 + f)
            with torch.no_grad():
                frames[prec] = nerf.run_one_iter_of_nerf(512, 512, bench.INTRINSICS, mc, mf, ro, rd, opt, mode="validation", encode_position_fn=ex,
                                                         encode_direction_fn=ed, expressions=expr, background_prior=bg, latent_code=lat)[3]
        for prec in PRECS:
            r = G.gate_cells(frames["f32"], frames[prec], seed=11 + f)
            r["frame"] = f
            rows[prec].append(r)
        print(f"[{scene}] frame {f}: " + " | ".join(f"{p} self {rows[p][-1]['self_psnr_db']:.1f} dB, 30dB whole {rows[p][-1]['cells']['30dB']['whole']:.1e}" for p in PRECS), flush=True)
    result["scenes"][scene] = {p: {"worst": G.worst_of(rows[p]), "per_frame": rows[p]} for p in PRECS}
nerf.set_mlp_precision("f32")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
json.dump(result, open(out_path, "w"))
cols = [(t, m) for t in ("random", "20dB", "30dB", "40dB") for m in ("whole", "3001", "1024")]
print("\n| scene | arithmetic | self-PSNR dB | " + " | ".join(f"{t} {m}" for t, m in cols) + " |")
print("|---|---|---|" + "---|" * len(cols))
for scene, per in result["scenes"].items():
    for p in PRECS:
        w = per[p]["worst"]
        print(f"| {scene} | {p} | {w['min_self_psnr_db']:.1f} .. {w['max_self_psnr_db']:.1f} | " +
              " | ".join(("**%.1e**" if w["cells"][t][m] > G.GATE_DB else "%.1e") % w["cells"][t][m] for t, m in cols) + " |")
print(f"(worst over {n} frames per scene and 8 subsets per size; bold = above the 1e-4 dB gate)")
