"""Per-kernel times of the exact-f32 training kernels of the paper model (forward with saves | dX chain | weight-gradient GEMMs) at the
two launch sizes of a training iteration, HIP events on the launch stream (the backward through nf_paper_mlp_bwd_stage_ms).
NERFACE_HIP_LIB selects an experiment build.    python tools/time_train_f32.py [reps]"""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from nerf import _hip as H  # noqa: E402
from nerf import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0")
lib = H.lib()
m = bench.synth_params(1, dev)
hw = m.hip_weights()
packed, packed_t = hw.get(), hw.get_t()
g = torch.Generator(device="cpu").manual_seed(5)
n_rays = 2048
ro = torch.zeros(n_rays, 3).to(dev)
rd = (torch.randn(n_rays, 3, generator=g) * 0.3).to(dev)
cond = ops.paper_condition(packed, (torch.randn(76, generator=g) * 0.5).to(dev), (torch.randn(32, generator=g) * 0.1).to(dev), bench.NEAR, bench.FAR)
flat = torch.empty(lib.nf_paper_grad_floats(), device=dev)
tot = 0.0
for S in (64, 128):
    n = n_rays * S
    z = torch.sort(torch.rand(n_rays, S, generator=g) * 0.6 + 0.2, dim=-1)[0].to(dev)
    d_raw = (torch.randn(n_rays, S, 4, generator=g) / (3 * n_rays)).to(dev)
    raw = torch.empty(n_rays, S, 4, device=dev)
    saved = torch.empty(lib.nf_paper_saved_floats(n), device=dev)
    wsf = lib.nf_paper_bwd_workspace_floats(n)
    ws = torch.empty(wsf, device=dev)
    ms = [0.0] * 4
    for it in range(reps + 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        H.check(lib.nf_paper_mlp_fwd_train(H.ptr(packed), H.ptr(cond), H.ptr(ro), H.ptr(rd), H.ptr(rd), H.ptr(z), n_rays, S, H.ptr(raw), H.ptr(saved),
                                           H.stream_ptr(dev)), "fwd")
        e1.record()
        st = (C.c_float * 3)()
        H.check(lib.nf_paper_mlp_bwd_stage_ms(H.ptr(packed), H.ptr(packed_t), 0, H.ptr(cond), H.ptr(saved), H.ptr(d_raw), n_rays, S, H.ptr(ws), wsf,
                                              H.ptr(flat), st, H.stream_ptr(dev)), "bwd")
        if it >= 2:
            ms[0] += e0.elapsed_time(e1) / reps
            for k in range(3):
                ms[k + 1] += st[k] / reps
    tot += sum(ms)
    print(f"{n_rays}x{S}: fwd+saves {ms[0]:.3f}  chain {ms[1]:.3f}  dW {ms[2]:.3f}  reduce+unpack {ms[3]:.3f} ms")
print(f"MLP kernels of one iteration: {tot:.3f} ms   ({os.environ.get('NERFACE_HIP_LIB', 'default lib')})")
