"""Fingerprint of the exact-f32 backward of the paper model (dZ sections in the workspace + the 26 gradients, four sizes incl. ragged ones)
and its stage times.  Used for same-box A/B of kernel variants (NERFACE_HIP_LIB): a restructured chain must leave every hash unchanged
(profiles/r04_experiments.md section 8)."""
import argparse, hashlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
from nerf import ops, _hip as HH
dev = torch.device("cuda:0")
lib = HH.lib()
m = bench.synth_params(1, dev).train()
hw = m.hip_weights()
g = torch.Generator().manual_seed(11)
expr, lat = (torch.randn(76, generator=g) * 0.5).to(dev), (torch.randn(32, generator=g) * 0.1).to(dev)
cond = ops.paper_condition(hw.get(), expr, lat, 0.2, 0.8)
for n_rays, s in ((3, 7), (37, 128), (2047, 127), (2048, 128)):
    ro = torch.zeros(n_rays, 3, device=dev); rd = (torch.randn(n_rays, 3, generator=g) * 0.3).to(dev)
    z = torch.sort(torch.rand(n_rays, s, generator=g) * 0.6 + 0.2, dim=-1)[0].to(dev)
    d_raw = (torch.randn(n_rays, s, 4, generator=g) / (3 * n_rays)).to(dev)
    raw, (saved,) = ops.paper_mlp_fwd_train(hw.get(), cond, ro, rd, z, rd)
    n = n_rays * s
    ws_floats = lib.nf_paper_bwd_workspace_floats(n)
    ws = torch.zeros(ws_floats, device=dev)
    flat = torch.empty(lib.nf_paper_grad_floats(), device=dev)
    HH.check(lib.nf_paper_mlp_bwd(HH.ptr(hw.get()), HH.ptr(hw.get_t()), HH.ptr(cond), HH.ptr(saved), HH.ptr(d_raw), n_rays, s, HH.ptr(ws), ws_floats,
                                  HH.ptr(flat), HH.stream_ptr(dev)), "bwd")
    torch.cuda.synchronize()
    hz = hashlib.sha1(ws[:2176 * n].cpu().numpy().tobytes()).hexdigest()[:16]
    hg = hashlib.sha1(flat.cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"{n_rays}x{s}: dz sha1 {hz}  grads sha1 {hg}  finite {bool(torch.isfinite(flat).all())}", flush=True)
r = bench.train_roofline(argparse.Namespace(precision="f32", family="paper"), m, dev, 2048)
ks = r["kernels"]
print(f"f32 @262144: fwd_save {ks[0]['avg_launch_ms']:.3f}  chain {ks[1]['avg_launch_ms']:.3f}  dw {ks[2]['avg_launch_ms']:.3f} | @131072: chain {ks[1]['avg_launch_ms_64_samples']:.3f}", flush=True)
