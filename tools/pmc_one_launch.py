"""Three launches of the fused MLP forward at the bench's fine-pass shape (65536 rays x 192 samples), for the rocprofv3 PMC
passes bench.py runs (`roofline.traffic`).  argv[1]: f32 | bf16x3."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
import nerf  # noqa: E402
from nerf import ops  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
dev = torch.device("cuda:0")
m = bench.synth_params(1, dev)
hw = m.hip_weights()
cond = ops.paper_condition(hw.get(), torch.randn(76, device=dev) * 0.5, torch.randn(32, device=dev) * 0.1, bench.NEAR, bench.FAR)
ro, rd = nerf.get_ray_bundle(512, 512, bench.INTRINSICS, bench.frame_pose(0).to(dev))
n_rays, S = bench.CHUNK, bench.N_COARSE + bench.N_FINE
ro, rd = ro.view(-1, 3)[:n_rays].contiguous(), rd.view(-1, 3)[:n_rays].contiguous()
z = torch.sort(torch.rand((n_rays, S), device=dev) * 0.6 + 0.2, dim=-1)[0].contiguous()
for _ in range(3):
    raw = (ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, ro, rd, z) if prec == "bf16x3" else
           ops.paper_mlp_fwd_f16(hw.get_f16(), cond, ro, rd, z) if prec == "f16x3" else
           ops.paper_mlp_fwd_f16x2(hw.get_f16(), cond, ro, rd, z) if prec == "f16x2" else ops.paper_mlp_fwd(hw.get(), cond, ro, rd, z))
torch.cuda.synchronize()
print("pmc_one_launch", prec, float(raw[0, 0, 3]))
