"""Fingerprints and launch times of the split-fp16 forward kernels, both model families, inference and training forward -- for same-box
A/B of kernel variants (run once per library: NERFACE_HIP_LIB=<variant .so>): a restructured epilogue must leave every hash unchanged.
raw outputs at three shapes (ragged ones included), the training forward's raw + saved sections, then HIP-event times at the bench's
fine-pass shape (65536 rays x 192 samples) and at the training shape (2048 x 128).  argv: "notime" skips the timing."""
import hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
from nerf import ops

dev = torch.device("cuda:0")
sha = lambda t: hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:16]
print("library:", os.environ.get("NERFACE_HIP_LIB", "default"), flush=True)
g = torch.Generator().manual_seed(21)
expr, lat = (torch.randn(76, generator=g) * 0.5).to(dev), (torch.randn(32, generator=g) * 0.1).to(dev)
ro_all, rd_all = nerf.get_ray_bundle(512, 512, bench.INTRINSICS, bench.frame_pose(0).to(dev))
ro_all, rd_all = ro_all.view(-1, 3), rd_all.view(-1, 3)


def flat(o):
    if torch.is_tensor(o):
        return [o]
    return [t for e in o for t in flat(e)] if isinstance(o, (tuple, list)) else []


def inputs(n_rays, s):
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0].to(dev).contiguous()
    return ro_all[:n_rays].contiguous(), rd_all[:n_rays].contiguous(), z


for fam in ("paper", "lcode"):
    m = bench.synth_params(1, dev, fam)
    for n_rays, s in ((1000, 37), (4096, 64), (3, 5)):
        ro, rd, z = inputs(n_rays, s)
        for prec in ("f16x3", "f16x2", "bf16x3"):
            nerf.set_mlp_precision(prec)
            with torch.no_grad():
                raw, _ = m.hip_forward(ro, rd, z, rd, expr, lat, bench.NEAR, bench.FAR, False)
            print(f"hash {fam} {prec} fwd {n_rays}x{s}: {sha(raw)}  finite {bool(torch.isfinite(raw).all())}", flush=True)
        for prec in ("f16x3", "bf16x3"):
            nerf.set_mlp_precision(prec)
            raw, state = m.hip_forward(ro, rd, z, rd, expr, lat, bench.NEAR, bench.FAR, True)
            saved = [t for t in flat(state) if t.numel() > 4096]
            print(f"hash {fam} {prec} train fwd {n_rays}x{s}: raw {sha(raw)} saved " + " ".join(sha(t) for t in saved), flush=True)
    if "notime" in sys.argv:
        continue
    for n_rays, s, train in ((65536, 192, False), (65536, 64, False), (2048, 128, True)):
        ro, rd, z = inputs(n_rays, s)
        for prec in ("f16x3", "f16x2", "bf16x3"):
            if train and prec == "f16x2":
                continue
            nerf.set_mlp_precision(prec)
            fn = lambda: m.hip_forward(ro, rd, z, rd, expr, lat, bench.NEAR, bench.FAR, train)
            with torch.set_grad_enabled(train):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                reps = 40 if train else 8
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
            print(f"time {fam} {prec} {'train fwd' if train else 'fwd'} {n_rays}x{s}: {e0.elapsed_time(e1) / reps:.3f} ms", flush=True)
nerf.set_mlp_precision("f32")
