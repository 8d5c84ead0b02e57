"""Diagnostic: gradients of the three training arithmetics at BASELINE training size against an fp64 autograd evaluation of the
oracle MLP on the device, each at the ReLU masks its own forward saved."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, nerf
from nerf import ops
from oracle import cases as C, nerface_oracle as O
from tests import util as U
from tests.test_gpu_backward import SAVED, RELU_ORDER, saved_section, rel_l2
gpu = torch.device("cuda:0")
c = C.build_case("train_rand_64_64")
n_rays, s = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 128
g = torch.Generator().manual_seed(29)
ro, rd, _, _, _ = C.ray_subset(512, 512, 9, n_rays, 29)
z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0]
d_raw = torch.randn((n_rays, s, 4), generator=g) * (1.0 / (n_rays * 3))
p = c["p_fine"]
m = U.make_model(nerf, p, gpu)
hw = m.hip_weights(); pk = hw.get()
cond = ops.paper_condition(pk, c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
dv = lambda t: t.to(gpu)
n_pts = n_rays * s
for mode in ("f32", "f16", "bf16"):
    _, saved = ops.paper_mlp_fwd_train(pk, cond, dv(ro), dv(rd), dv(z), packed_b=hw.get_bf16() if mode == "bf16" else None,
                                       packed_h=hw.get_f16() if mode == "f16" else None)
    grads, g_lat = ops.paper_mlp_bwd(m, pk, cond, dv(z), dv(d_raw), saved, split={"f32": False, "f16": "f16", "bf16": True}[mode])
    from tests.test_gpu_backward import f32_rows
    sv = f32_rows(saved[0], n_pts, mode)
    masks = [saved_section(sv, k, n_pts) > 0 for k in RELU_ORDER]
    pp = {k: v.to(gpu).double().clone().requires_grad_(True) for k, v in p.items()}
    lat = c["latent"].to(gpu).double().clone().requires_grad_(True)
    x = O.encode_points(dv(ro).double(), dv(rd).double(), dv(z).double(), O.NEAR, O.FAR)
    out = O.paper_mlp(pp, x, c["expr"].to(gpu).double(), lat, masks=masks)
    out.backward(dv(d_raw).reshape(-1, 4).double())
    errs = {k: rel_l2(gh, pp[k].grad) for k, gh in zip(ops.PAPER_KEYS, grads) if gh is not None}
    worst = max(errs, key=errs.get)
    print(f"{mode:5s} {n_rays}x{s}: worst {worst} {errs[worst]:.2e}  latent {rel_l2(g_lat, lat.grad):.2e}  fc_rgb.w {errs['fc_rgb.weight']:.2e} l_dir.2.w {errs['layers_dir.2.weight']:.2e} xyz.5.w {errs['layers_xyz.5.weight']:.2e} xyz.0.w {errs['layers_xyz.0.weight']:.2e} xyz.0.b {errs['layers_xyz.0.bias']:.2e}")
    del pp, out, x
