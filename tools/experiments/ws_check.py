"""Fingerprint + launch time of the split inference kernels of ONE process (the weight-stationary build reads NERFACE_SPLIT_WS once):
run it twice (0 / 1) and compare.  Belongs to the archived experiment nf_mlp_bf16_ws.inc (see README.md here)."""
import hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
from nerf import ops
dev = torch.device("cuda:0")
m = bench.synth_params(1, dev)
hw = m.hip_weights()
torch.manual_seed(3)
cond = ops.paper_condition(hw.get(), torch.randn(76, device=dev) * 0.5, torch.randn(32, device=dev) * 0.1, 0.2, 0.8)
ro, rd = nerf.get_ray_bundle(512, 512, bench.INTRINSICS, bench.frame_pose(0).to(dev))
for n_rays, S in ((1000, 37), (4096, 64), (3, 5), (65536, 192)):
    ro_, rd_ = ro.view(-1, 3)[:n_rays].contiguous(), rd.view(-1, 3)[:n_rays].contiguous()
    z = torch.sort(torch.rand((n_rays, S), device=dev) * 0.6 + 0.2, dim=-1)[0].contiguous()
    for name, fn in (("bf16x3", lambda: ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, ro_, rd_, z)), ("f16x3", lambda: ops.paper_mlp_fwd_f16(hw.get_f16(), cond, ro_, rd_, z))):
        raw = fn()
        torch.cuda.synchronize()
        line = f"{name} {n_rays}x{S}: sha1 {hashlib.sha1(raw.cpu().numpy().tobytes()).hexdigest()[:16]} finite {bool(torch.isfinite(raw).all())}"
        if n_rays == 65536:
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): fn()
            e1.record(); torch.cuda.synchronize()
            line += f"  {e0.elapsed_time(e1) / 5:.3f} ms per launch"
        print(line, flush=True)
