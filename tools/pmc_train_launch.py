"""Three launches of the training forward + backward of the paper model at the fine-pass training shape (2048 rays x 128 samples) in
each arithmetic named on the command line, for the rocprofv3 PMC passes bench.py runs (`train.<prec>.roofline.kernels[].traffic`).
argv[1:]: any of f32 | bf16x3 | f16x3."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from nerf import ops  # noqa: E402

dev = torch.device("cuda:0")
m = bench.synth_params(1, dev)
hw = m.hip_weights()
g = torch.Generator(device="cpu").manual_seed(5)
n_rays, S = 2048, 128
ro = torch.zeros(n_rays, 3).to(dev)
rd = (torch.randn(n_rays, 3, generator=g) * 0.3).to(dev)
expr, lat = (torch.randn(76, generator=g) * 0.5).to(dev), (torch.randn(32, generator=g) * 0.1).to(dev)
z = torch.sort(torch.rand(n_rays, S, generator=g) * 0.6 + 0.2, dim=-1)[0].to(dev)
d_raw = (torch.randn(n_rays, S, 4, generator=g) / (3 * n_rays)).to(dev)
cond = ops.paper_condition(hw.get(), expr, lat, bench.NEAR, bench.FAR)
for prec in (sys.argv[1:] or ["f32"]):
    for _ in range(3):
        raw, saved = ops.paper_mlp_fwd_train(hw.get(), cond, ro, rd, z, rd, packed_b=hw.get_bf16() if prec == "bf16x3" else None,
                                             packed_h=hw.get_f16() if prec == "f16x3" else None)
        grads, g_lat = ops.paper_mlp_bwd(m, hw.get(), cond, z, d_raw, saved, split={"f32": False, "f16x3": "f16", "bf16x3": True}[prec])
    torch.cuda.synchronize()
    print("pmc_train_launch", prec, float(g_lat[0]))
