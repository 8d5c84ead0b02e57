"""Training iteration of bench.py (configs[2]) with the fused loss (nerf.training_loss) against the trainer's torch expression, same
process, alternating, per arithmetic."""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
dev = torch.device("cuda:0")
fused = nerf.training_loss


def torch_loss(rc, rf, tgt, lat):
    return torch.nn.functional.mse_loss(rc, tgt) + torch.nn.functional.mse_loss(rf, tgt) + 10 * 0.0005 * torch.norm(lat), None


precs = sys.argv[1:] or ["f32", "bf16x3", "f16x3"]
for prec in precs:
    nerf.set_mlp_precision(prec)
    for rep in range(2):
        for name, fn in (("torch loss", torch_loss), ("fused loss", fused)):
            nerf.training_loss = fn
            args = argparse.Namespace(steps=60, warmup=8, precision=prec, family="paper", gpus=1)
            bench.train_roofline = lambda *a, **k: None          # (only the iteration time is wanted here)
            r = bench.bench_train(args, nerf, bench.synth_params(0, dev, "paper"), bench.synth_params(1, dev, "paper"), dev, 0, 1, None, emit=False)
            print(f"{prec} {name}: {r['ms_per_step']:.3f} ms per iteration", flush=True)
