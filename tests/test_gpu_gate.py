"""north_star's gate where it is hard: |PSNR(arithmetic, target) - PSNR(exact f32, target)| <= 1e-4 dB against targets the
render approximates to 20 / 30 / 40 dB (not only the uniform-random target of SURVEY 8(d), which is ~170x less sensitive) and on scattered
subsets of 1,024 and 3,001 rays beside whole 512 x 512 frames -- every inference arithmetic, eight frames of bench.py's scene (x1000 density
head) and four of the same scene with SURVEY 8(d)'s x40 head.  What each arithmetic passes is an explicit expectation (nerf.gate.EXPECTED_PASS,
from profiles/r06_gate_sensitivity.md): the test fails when a cell expected to pass misses the gate AND when a cell recorded as failing
starts to pass by more than 2x (the claim in include/nerface_hip.h / README would then be stale).  The fp32 noise floor of the same cells --
the product's exact-f32 frame against the ORACLE evaluated in float64 -- is measured beside it: no fp32 implementation, the reference's own
included, can be closer to another one than that.  GPU only."""
import json
import os

import pytest
import torch

from oracle import nerface_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu
PRECS = ("f16x3", "bf16x3", "f16x2")
CELLS = [(t, m) for t in ("random", "20dB", "30dB", "40dB") for m in ("whole", "3001", "1024")]


def _scene_models(nerf, bench, gpu, scene):
    mc, mf = bench.synth_params(0, gpu), bench.synth_params(1, gpu)
    if scene == "soft":
        with torch.no_grad():
            for m in (mc, mf):
                m.fc_alpha.weight.mul_(40.0 / 1000.0)
                m.fc_alpha.bias.fill_(0.5)
    return mc, mf


def _cond(f):
    """bench.py's per-frame expression / latent code (its own float32 draws)."""
    g = torch.Generator().manual_seed(1000 + f)
    return 0.5 * torch.randn(76, generator=g), 0.1 * torch.randn(32, generator=g)


def _render(nerf, bench, mc, mf, gpu, f, prec, rands=None):
    opt = bench.options(nerf)
    ex, ed = U.encoders(nerf)
    bg = torch.rand((512, 512, 3), generator=torch.Generator().manual_seed(7)).to(gpu).view(-1, 3)
    expr, lat = _cond(f)
    ro, rd = nerf.get_ray_bundle(512, 512, bench.INTRINSICS, bench.frame_pose(f).to(gpu))
    nerf.set_mlp_precision(prec)
    try:
        torch.manual_seed(4321 + f)
        with torch.no_grad():
            if rands is not None:
                with U.injected_random(rands, []):
                    return nerf.run_one_iter_of_nerf(512, 512, bench.INTRINSICS, mc, mf, ro, rd, opt, mode="validation", encode_position_fn=ex,
                                                     encode_direction_fn=ed, expressions=expr.to(gpu), background_prior=bg, latent_code=lat.to(gpu))[3]
            return nerf.run_one_iter_of_nerf(512, 512, bench.INTRINSICS, mc, mf, ro, rd, opt, mode="validation", encode_position_fn=ex,
                                             encode_direction_fn=ed, expressions=expr.to(gpu), background_prior=bg, latent_code=lat.to(gpu))[3]
    finally:
        nerf.set_mlp_precision("f32")


def _fmt(w):
    return " | ".join(f"{t} {m} {w['cells'][t][m]:.1e}" for t, m in CELLS)


@pytest.mark.parametrize("scene,n_frames", [("bench", 8), ("soft", 4)])
def test_gate_sensitivity_of_every_arithmetic(hip_lib, gpu, scene, n_frames):
    import bench
    import nerf
    from nerf import gate as G
    mc, mf = _scene_models(nerf, bench, gpu, scene)
    rows = {p: [] for p in PRECS}
    for f in range(n_frames):
        exact = _render(nerf, bench, mc, mf, gpu, f, "f32")
        for p in PRECS:
            rows[p].append(G.gate_cells(exact, _render(nerf, bench, mc, mf, gpu, f, p), seed=11 + f))
    worst = {p: G.worst_of(rows[p]) for p in PRECS}
    for p in PRECS:
        print(f"[{scene}] {p}: self-PSNR {worst[p]['min_self_psnr_db']:.1f} .. {worst[p]['max_self_psnr_db']:.1f} dB | {_fmt(worst[p])}")
    dump = os.environ.get("NERFACE_GATE_JSON")
    if dump:
        with open(f"{dump}.{scene}.json", "w") as fh:
            json.dump({p: {"worst": worst[p], "per_frame": rows[p]} for p in PRECS}, fh)
    if os.environ.get("NERFACE_GATE_MEASURE"):
        return
    bad = []
    for p in PRECS:
        for t, m in CELLS:
            v, want = worst[p]["cells"][t][m], G.EXPECTED_PASS[p][scene][t][m]
            if want is True and v > G.GATE_DB:
                bad.append((p, t, m, v, "expected to pass"))
            if want is False and v <= 0.5 * G.GATE_DB:
                bad.append((p, t, m, v, "recorded as failing, now passes with margin: restate the claim"))
    assert not bad, bad


@pytest.mark.parametrize("scene", ["bench", "soft"])
def test_fp32_noise_floor_of_the_gate(hip_lib, gpu, scene):
    """The same cells for the exact-f32 product against the oracle in float64 on the device (same stratified jitter and inverse-CDF
    abscissae on both sides, fed as tensors), two frames per scene: the distance any two fp32 implementations of this path keep from
    each other.  f32 must pass every whole-frame cell.  "fp32-class" made checkable: on the same frames and draws, f16x3's distance to
    the f32 frame stays within max(gate, 2 x the f32 frame's own distance to float64) in EVERY cell -- where f16x3 misses the gate
    (x1000 head, small ray sets, 30 / 40 dB targets) the exact-f32 arithmetic misses it too."""
    import bench
    import nerf
    from nerf import gate as G
    mc, mf = _scene_models(nerf, bench, gpu, scene)
    pc = {k: v.detach().cpu() for k, v in mc.state_dict().items()}
    pf = {k: v.detach().cpu() for k, v in mf.state_dict().items()}
    rows, rows16 = [], []
    for f in range(2):
        g = torch.Generator().manual_seed(20260930 + f)
        t_rand, u = torch.rand((512 * 512, 64), generator=g), torch.rand((512 * 512, 128), generator=g)
        rands = [t for k in range(0, 512 * 512, bench.CHUNK) for t in (t_rand[k:k + bench.CHUNK], u[k:k + bench.CHUNK])]
        ours = _render(nerf, bench, mc, mf, gpu, f, "f32", rands=rands)
        ours16 = _render(nerf, bench, mc, mf, gpu, f, "f16x3", rands=rands)
        expr, lat = _cond(f)
        ro_c, rd_c = O.ray_bundle(512, 512, bench.INTRINSICS, bench.frame_pose(f))
        bg = torch.rand((512, 512, 3), generator=torch.Generator().manual_seed(7)).view(-1, 3)
        c = dict(p_coarse=pc, p_fine=pf, expr=expr, latent=lat)
        ref = U.oracle_render_fp64_on_device(c, ro_c.reshape(-1, 3), rd_c.reshape(-1, 3), bg, gpu, 64, 128, t_rand=t_rand, u=u)[3]
        rows.append(G.gate_cells(ref, ours, seed=11 + f))
        rows16.append(G.gate_cells(ours, ours16, seed=11 + f))
    w, w16 = G.worst_of(rows), G.worst_of(rows16)
    print(f"[{scene}] f32 vs fp64 oracle: self-PSNR {w['min_self_psnr_db']:.1f} .. {w['max_self_psnr_db']:.1f} dB | {_fmt(w)}")
    print(f"[{scene}] f16x3 vs f32, same frames and draws: self-PSNR {w16['min_self_psnr_db']:.1f} .. {w16['max_self_psnr_db']:.1f} dB | {_fmt(w16)}")
    dump = os.environ.get("NERFACE_GATE_JSON")
    if dump:
        with open(f"{dump}.{scene}.floor.json", "w") as fh:
            json.dump({"worst": w, "per_frame": rows, "f16x3_same_draws": w16}, fh)
    for t in ("random", "20dB", "30dB", "40dB"):
        assert w["cells"][t]["whole"] <= G.GATE_DB, (t, w["cells"][t]["whole"])
    bad = [(t, m, w16["cells"][t][m], w["cells"][t][m]) for t, m in CELLS if w16["cells"][t][m] > max(G.GATE_DB, 2.0 * w["cells"][t][m])]
    assert not bad, bad
