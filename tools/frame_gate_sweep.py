"""north_star's gate over MANY frames of the bench workload: |PSNR(arithmetic, target) - PSNR(exact f32, target)| and the self-PSNR of the
whole 512 x 512 frame for f16x3 / f16x2 / bf16x3 against the product's exact-f32 frame (same seeded draws), frames 0..N-1 of bench.py's scene
(its poses, expressions, latent codes) with per-frame random targets.  The metric varies by an order of magnitude from scene to scene
(profiles/r05_split_products.md): one frame is not a margin.  argv: number of frames (default 12)."""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
mc, mf = bench.synth_params(0, dev), bench.synth_params(1, dev)
opt = bench.options(nerf)
ex = nerf.get_embedding_function(num_encoding_functions=10, include_input=True, log_sampling=True)
ed = nerf.get_embedding_function(num_encoding_functions=4, include_input=False, log_sampling=True)
bg = torch.rand((512, 512, 3), generator=torch.Generator().manual_seed(7)).to(dev).view(-1, 3)
psnr = lambda a, b: -10.0 * math.log10(float(((a - b) ** 2).mean()))
worst = {}
for f in range(n):
    g = torch.Generator().manual_seed(1000 + f)
    expr, lat = (0.5 * torch.randn(76, generator=g)).to(dev), (0.1 * torch.randn(32, generator=g)).to(dev)
    tgt = torch.rand((512, 512, 3), generator=torch.Generator().manual_seed(11 + f)).to(dev).double()
    ro, rd = nerf.get_ray_bundle(512, 512, bench.INTRINSICS, bench.frame_pose(f).to(dev))
    frames = {}
    for prec in ("f32", "f16x3", "f16x2", "bf16x3"):
        nerf.set_mlp_precision(prec)
        torch.manual_seed(4321 + f)
        with torch.no_grad():
            frames[prec] = nerf.run_one_iter_of_nerf(512, 512, bench.INTRINSICS, mc, mf, ro, rd, opt, mode="validation", encode_position_fn=ex,
                                                     encode_direction_fn=ed, expressions=expr, background_prior=bg, latent_code=lat)[3].double()
    row = []
    for prec in ("f16x3", "f16x2", "bf16x3"):
        dp, sp = abs(psnr(frames[prec], tgt) - psnr(frames["f32"], tgt)), psnr(frames[prec], frames["f32"])
        worst[prec] = max(worst.get(prec, 0.0), dp)
        row.append(f"{prec} {dp:.2e} dB ({sp:.1f} dB)")
    print(f"frame {f:2d}: " + " | ".join(row), flush=True)
nerf.set_mlp_precision("f32")
print("worst |dPSNR| over", n, "frames:", {k: f"{v:.2e}" for k, v in worst.items()}, "(gate 1e-4)")
