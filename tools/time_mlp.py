"""Time the fused MLP forward kernels (exact f32 vs split bf16) at the bench shape with HIP events."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
from nerf import ops
dev = torch.device("cuda:0")
m = bench.synth_params(1, dev)
hw = m.hip_weights()
cond = ops.paper_condition(hw.get(), torch.randn(76, device=dev) * 0.5, torch.randn(32, device=dev) * 0.1, 0.2, 0.8)
ro, rd = nerf.get_ray_bundle(512, 512, bench.INTRINSICS, bench.frame_pose(0).to(dev))
if os.environ.get("TIME_MLP_HASH"):                # bit-level fingerprint of the f32 forward (A/B of kernel variants across processes)
    import hashlib
    torch.manual_seed(7)
    for n_rays, S in ((1000, 37), (4096, 64), (3, 5)):
        ro_, rd_ = ro.view(-1, 3)[:n_rays].contiguous(), rd.view(-1, 3)[:n_rays].contiguous()
        z = torch.sort(torch.rand((n_rays, S), device=dev) * 0.6 + 0.2, dim=-1)[0].contiguous()
        raw = ops.paper_mlp_fwd(hw.get(), cond, ro_, rd_, z)
        print(f"hash f32 {n_rays}x{S}: {hashlib.sha1(raw.cpu().numpy().tobytes()).hexdigest()}  sum {raw.double().sum().item():.9e}  finite {bool(torch.isfinite(raw).all())}")
for n_rays, S in ((65536, 192), (65536, 64)):
    ro_, rd_ = ro.view(-1, 3)[:n_rays].contiguous(), rd.view(-1, 3)[:n_rays].contiguous()
    z = torch.sort(torch.rand((n_rays, S), device=dev) * 0.6 + 0.2, dim=-1)[0].contiguous()
    for name, fn in (("f32", lambda: ops.paper_mlp_fwd(hw.get(), cond, ro_, rd_, z)), ("bf16x3", lambda: ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, ro_, rd_, z)),
                     ("f16x3", lambda: ops.paper_mlp_fwd_f16(hw.get_f16(), cond, ro_, rd_, z)),
                     ("f16x2", lambda: ops.paper_mlp_fwd_f16x2(hw.get_f16(), cond, ro_, rd_, z))):
        if name == "f32" and os.environ.get("TIME_MLP_SKIP_F32"):
            continue
        if name != "f32" and os.environ.get("TIME_MLP_ONLY_F32"):
            continue
        if os.environ.get("TIME_MLP_ONLY") and name != os.environ["TIME_MLP_ONLY"]:
            continue
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tf = n_rays * S * 1100032 / (ms * 1e-3) / 1e12
        print(f"{name:7s} {n_rays}x{S}: {ms:8.3f} ms  {tf:7.1f} TFLOP/s algorithmic")
