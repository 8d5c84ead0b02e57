"""Per-kernel times of the training MLP kernels (forward with saves, dX chain, weight-gradient GEMMs, reduce) at the two launch sizes of
an iteration, through bench.train_roofline (HIP events on the launch stream).  argv: precisions (default: bf16x3 f16x3) [lcode]."""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
dev = torch.device("cuda:0")
precs = [a for a in sys.argv[1:] if a in ("f32", "bf16x3", "f16x3")] or ["bf16x3", "f16x3"]
fams = ["paper"] + (["lcode"] if "lcode" in sys.argv[1:] else [])
info = bench.device_info(dev)
print(f"box: hbm_fill {info.get('hbm_fill_gbs', 0):.0f} GB/s, hbm_copy {info.get('hbm_copy_gbs', 0):.0f} GB/s", flush=True)
for fam in fams:
    model = bench.synth_params(1, dev, fam).train()
    for prec in precs:
        nerf.set_mlp_precision(prec)
        r = bench.train_roofline(argparse.Namespace(precision=prec, family=fam), model, dev, 2048)
        if fam == "paper":
            ks = r["kernels"]
            print(f"{fam} {prec:7s} @262144: fwd_save {ks[0]['avg_launch_ms']:.3f}  chain {ks[1]['avg_launch_ms']:.3f}  dw {ks[2]['avg_launch_ms']:.3f}  | "
                  f"@131072: fwd_save {ks[0]['avg_launch_ms_64_samples']:.3f}  chain {ks[1]['avg_launch_ms_64_samples']:.3f}  dw {ks[2]['avg_launch_ms_64_samples']:.3f}  | "
                  f"MLP kernels of an iteration {r['ms_both_launches']:.3f} ms", flush=True)
        else:
            print(f"{fam} {prec:7s}: MLP kernels of an iteration (fwd+bwd, both launches) {r['ms_both_launches']:.3f} ms", flush=True)
