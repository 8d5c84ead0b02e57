# GPU call 22 (round 3): which `saved` section of the streamed training forward differs from the split-bf16 forward's
cd $GRAFT_REPO_ROOT
timeout 600 python - <<'PY' 2>&1 | grep -v Warning | tail -40
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "4d-facial-avatars_amd")
import torch
from tests import util as U
from tests import test_gpu_backward as T
import nerf
from nerf import ops
C, O = T.C, T.O
gpu = torch.device("cuda:0")
c = C.build_case("train_rand_64_64")
for n_rays, s in ((8, 64), (3, 7)):
    g = torch.Generator().manual_seed(17)
    ro, rd, _, _, _ = C.ray_subset(512, 512, 9, n_rays, 17)
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0]
    m = U.make_model(nerf, c["p_fine"], gpu)
    hw = m.hip_weights()
    cond = ops.paper_condition(hw.get(), c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    raw_f, (sv_f,) = ops.paper_mlp_fwd_train(hw.get(), cond, ro.to(gpu), rd.to(gpu), z.to(gpu))
    raw_b, (sv_b,) = ops.paper_mlp_fwd_train(hw.get(), cond, ro.to(gpu), rd.to(gpu), z.to(gpu), packed_b=hw.get_bf16())
    n = n_rays * s
    for name in T.SAVED:
        a, b = T.saved_section(sv_f.cpu(), name, n), T.saved_section(sv_b.cpu(), name, n)
        d = (a - b).abs()
        bad = (d > 3e-4 * (1 + float(a.abs().max()))).nonzero()
        print(n_rays, s, name, "max diff", float(d.max()), "n_bad", bad.shape[0], "rows", sorted(set(bad[:, 0].tolist()))[:12], "cols", sorted(set(bad[:, 1].tolist()))[:12], "neg in f32", int((a < 0).sum()))
    print("raw diff", float((raw_f - raw_b).abs().max()))
PY
