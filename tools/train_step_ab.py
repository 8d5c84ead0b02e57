"""Training iteration of bench.py (configs[2]) with the fused loss (nerf.training_loss, two launches) against the trainer's torch expression
(TR:355-387, ~20 launches), same process, alternating, per arithmetic.  (Round 4, call 25 also timed the ray selection drawn one iteration
ahead on a side stream: -17 us f32, nothing in the split arithmetics -- not kept; profiles/r04_experiments.md section 11.)"""
import argparse, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
dev = torch.device("cuda:0")
fused = nerf.training_loss


def torch_loss(rc, rf, tgt, lat):
    return torch.nn.functional.mse_loss(rc, tgt) + torch.nn.functional.mse_loss(rf, tgt) + 10 * 0.0005 * torch.norm(lat), None


bench.train_roofline = lambda *a, **k: None          # (only the iteration time is wanted here)
precs = sys.argv[1:] or ["f32", "bf16x3", "f16x3"]
for prec in precs:
    nerf.set_mlp_precision(prec)
    for rep in range(2):
        for name, loss_fn in (("torch loss", torch_loss), ("fused loss", fused)):
            nerf.training_loss = loss_fn
            args = argparse.Namespace(steps=60, warmup=8, precision=prec, family="paper", gpus=1)
            r = bench.bench_train(args, nerf, bench.synth_params(0, dev, "paper"), bench.synth_params(1, dev, "paper"), dev, 0, 1, None, emit=False)
            print(f"{prec} {name}: {r['ms_per_step']:.3f} ms per iteration", flush=True)
