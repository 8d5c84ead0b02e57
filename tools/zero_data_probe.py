"""Are the split-arithmetic kernels power-limited?  The same launches (65536 rays x 192 samples, same instruction stream) on the bench's random
weights and on all-zero weights / inputs: MI355X_MICROARCH.md ("DVFS give-back") measured +19 % for a matrix kernel on zero-filled inputs --
less switching, a higher granted clock.  A kernel that is issue- or latency-bound does not care what its operands are."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
from nerf import ops
dev = torch.device("cuda:0")
ro, rd = nerf.get_ray_bundle(512, 512, bench.INTRINSICS, bench.frame_pose(0).to(dev))
n_rays, S = 65536, 192
ro_, rd_ = ro.view(-1, 3)[:n_rays].contiguous(), rd.view(-1, 3)[:n_rays].contiguous()
for rep in range(2):
    for data in ("random", "zero"):
        m = bench.synth_params(1, dev)
        expr, lat = torch.randn(76, device=dev) * 0.5, torch.randn(32, device=dev) * 0.1
        z = torch.sort(torch.rand((n_rays, S), device=dev) * 0.6 + 0.2, dim=-1)[0].contiguous()
        ro_u, rd_u = ro_, rd_
        if data == "zero":
            with torch.no_grad():
                for p in m.parameters():
                    p.zero_()
            expr, lat, z = torch.zeros_like(expr), torch.zeros_like(lat), torch.zeros_like(z)
            ro_u, rd_u = torch.zeros_like(ro_), torch.zeros_like(rd_)
        hw = m.hip_weights()
        cond = ops.paper_condition(hw.get(), expr, lat, 0.2, 0.8)
        for name, fn in (("f32", lambda: ops.paper_mlp_fwd(hw.get(), cond, ro_u, rd_u, z)),
                         ("bf16x3", lambda: ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, ro_u, rd_u, z)),
                         ("f16x3", lambda: ops.paper_mlp_fwd_f16(hw.get_f16(), cond, ro_u, rd_u, z)),
                         ("f16x2", lambda: ops.paper_mlp_fwd_f16x2(hw.get_f16(), cond, ro_u, rd_u, z))):
            fn(); fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): fn()
            e1.record(); torch.cuda.synchronize()
            print(f"{name:7s} {data:6s} operands: {e0.elapsed_time(e1) / 8:8.3f} ms per 65536 x 192 launch", flush=True)
