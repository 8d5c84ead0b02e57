"""Diagnostic: the backward chains on ONE set of saved activations / masks (the fp16 forward's, which carries the bit masks the split chains read): dZ sections and their
column sums against fp64 autograd on the device."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, nerf
from nerf import ops
from oracle import cases as C, nerface_oracle as O
from tests import util as U
from tests.test_gpu_backward import RELU_ORDER, saved_section, rel_l2
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
_, saved = ops.paper_mlp_fwd_train(pk, cond, dv(ro), dv(rd), dv(z), packed_h=hw.get_f16())
from tests.test_gpu_backward import f32_rows
sv = f32_rows(saved[0], n_pts, "f16")
masks = [saved_section(sv, k, n_pts) > 0 for k in RELU_ORDER]
# fp64 reference with hooks on the pre-activation gradients we care about
pp = {k: v.to(gpu).double().clone().requires_grad_(True) for k, v in p.items()}
lat = c["latent"].to(gpu).double().clone().requires_grad_(True)
x = O.encode_points(dv(ro).double(), dv(rd).double(), dv(z).double(), O.NEAR, O.FAR)
out = O.paper_mlp(pp, x, c["expr"].to(gpu).double(), lat, masks=masks)
out.backward(dv(d_raw).reshape(-1, 4).double())
ref_bias = {k: pp[k].grad for k in pp if k.endswith("bias")}

ws_box = {}
real_empty = torch.empty
def spy(*a, **k):
    t = real_empty(*a, **k)
    if k.get("dtype") == torch.float32 and t.numel() > 10 ** 6:
        ws_box["ws"] = t
    return t
SEC = {"L0": (0, 256, "layers_xyz.0.bias"), "L1": (256, 256, "layers_xyz.1.bias"), "L5": (1280, 256, "layers_xyz.5.bias"),
       "FEAT": (1536, 256, "fc_feat.bias"), "D0": (1792, 128, "layers_dir.0.bias"), "D2": (2048, 128, "layers_dir.2.bias")}
dz = {}
for mode, kw in (("f32", dict(split=False)), ("bf16", dict(split=True)), ("f16", dict(split="f16"))):
    torch.empty = spy
    grads, g_lat = ops.paper_mlp_bwd(m, pk, cond, dv(z), dv(d_raw), saved, **kw)
    torch.empty = real_empty
    ws = ws_box["ws"]
    dz[mode] = {k: ws[o * n_pts:(o + w) * n_pts].view(n_pts, w).clone() for k, (o, w, _) in SEC.items()}
    errs = {k: rel_l2(gh, pp[k].grad) for k, gh in zip(ops.PAPER_KEYS, grads) if gh is not None}
    print(mode, "grads vs fp64: fc_feat.b %.2e xyz.0.b %.2e xyz.5.w %.2e l_dir.2.w %.2e latent %.2e" % (
        errs["fc_feat.bias"], errs["layers_xyz.0.bias"], errs["layers_xyz.5.weight"], errs["layers_dir.2.weight"], rel_l2(g_lat, lat.grad)))
    for k, (o, w, bk) in SEC.items():
        cs = dz[mode][k].double().sum(0)
        print(f"   {k:5s} fp64 column sums of the kernel's dz vs fp64 bias grad: {rel_l2(cs, ref_bias[bk]):.2e}   cancellation sum|dz|/|sum dz| {float(dz[mode][k].abs().sum() / cs.abs().sum()):.0f}")
for k in SEC:
    a, b, h = dz["f32"][k].double(), dz["bf16"][k].double(), dz["f16"][k].double()
    print(f"{k:5s} dz rel L2: bf16 vs f32 {rel_l2(b, a):.2e}   f16 vs f32 {rel_l2(h, a):.2e}   mean signed (f16-f32)/mean|f32| {float((h - a).mean() / a.abs().mean()):.2e}  (bf16-f32) {float((b - a).mean() / a.abs().mean()):.2e}")
