"""Split-fp16 ("f16x3") fused MLP forward: fp32-CLASS accuracy on the 16-bit matrix pipe.  GPU only.

The kernel is held to the gates of the EXACT-f32 kernel (not to the looser split-bf16 ones): raw outputs against an fp64
evaluation within the f32 kernel's tolerance and within a small factor of the f32 kernel's own error, the fine pass on the
oracle's depths at f32 tolerances, the PSNR gate, and robustness of the per-layer weight scales to extreme magnitudes."""
import os

import numpy as np
import pytest
import torch

from oracle import cases as C
from oracle import nerface_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu


def _raw_errors(nerf, ops, gpu, params, expr, latent, ro, rd, z):
    m = U.make_model(nerf, params, gpu)
    hw = m.hip_weights()
    cond = ops.paper_condition(hw.get(), expr.to(gpu), latent.to(gpu), O.NEAR, O.FAR)
    dv = lambda t: t.to(gpu).contiguous()
    p64 = {k: v.double() for k, v in params.items()}
    ref = O.paper_mlp(p64, O.encode_points(ro.double(), rd.double(), z.double(), O.NEAR, O.FAR), expr.double(), latent.double()).reshape(*z.shape, 4)
    out = {"f32": ops.paper_mlp_fwd(hw.get(), cond, dv(ro), dv(rd), dv(z)),
           "bf16x3": ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, dv(ro), dv(rd), dv(z)),
           "f16x3": ops.paper_mlp_fwd_f16(hw.get_f16(), cond, dv(ro), dv(rd), dv(z))}
    scale = ref.abs().amax(dim=(0, 1))
    err = {k: (v.cpu().double() - ref).abs().amax(dim=(0, 1)) for k, v in out.items()}
    rms = {k: (v.cpu().double() - ref).pow(2).mean(dim=(0, 1)).sqrt() for k, v in out.items()}
    return err, rms, scale, out


@pytest.mark.parametrize("boost", [True, "survey"])
def test_f16x3_raw_outputs_meet_the_f32_gate(hip_lib, gpu, boost):
    import nerf
    from nerf import ops
    c = C.build_case("eval_det_64_128")
    g = torch.Generator().manual_seed(5)
    n_rays, s = 96, 192
    ro, rd, _, _, _ = C.ray_subset(512, 512, 3, n_rays, 5)
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0]
    params = O.init_paper_params(1, boost=boost)
    err, rms, scale, out = _raw_errors(nerf, ops, gpu, params, c["expr"], c["latent"], ro, rd, z)
    for k in ("f32", "f16x3", "bf16x3"):
        print(f"[boost={boost}] {k:7s} max|err| vs fp64 {['%.2e' % v for v in err[k].tolist()]}  rms {['%.2e' % v for v in rms[k].tolist()]}  (scale {['%.2g' % v for v in scale.tolist()]})")
    assert torch.all(err["f16x3"] <= 2e-5 * scale + 2e-5)                       # the exact-f32 kernel's own gate (test_paper_mlp_fwd)
    assert torch.all(rms["f16x3"] <= 4.0 * rms["f32"] + 1e-9)                   # fp32-class: within a small factor of the f32 kernel's error
    assert torch.all(rms["f16x3"] <= 0.25 * rms["bf16x3"])                      # and far below the split-bf16 error
    again = ops.paper_mlp_fwd_f16(U.make_model(nerf, params, gpu).hip_weights().get_f16(),
                                  ops.paper_condition(U.make_model(nerf, params, gpu).hip_weights().get(), c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR),
                                  ro.to(gpu), rd.to(gpu), z.to(gpu))
    assert torch.equal(again, out["f16x3"])                                     # deterministic


def test_f16x3_weight_scales_follow_the_layers(hip_lib, gpu):
    """Per-layer power-of-two weight scales: layers whose weights differ by 2^12 in magnitude from their neighbours (x 2^6 in one
    layer, x 2^-6 in the next -- ReLU is positively homogeneous, the function is unchanged up to rounding; hidden activations
    range from O(0.01) to O(300)) keep the f32-gate accuracy."""
    import nerf
    from nerf import ops
    c = C.build_case("eval_det_64_128")
    params = O.init_paper_params(3, boost="survey")
    k = 2.0 ** 6
    for name, f in (("layers_xyz.1", k), ("layers_xyz.2", 1 / k), ("layers_xyz.4", 1 / k), ("layers_xyz.5", k), ("layers_dir.1", k), ("layers_dir.2", 1 / k)):
        params[name + ".weight"] = params[name + ".weight"] * f
    for name in ("layers_xyz.1", "layers_xyz.4", "layers_xyz.5", "layers_dir.1"):
        params[name + ".bias"] = params[name + ".bias"] * (k if name != "layers_xyz.4" else 1 / k)
    g = torch.Generator().manual_seed(6)
    ro, rd, _, _, _ = C.ray_subset(512, 512, 3, 32, 6)
    z = torch.sort(torch.rand((32, 64), generator=g) * 0.6 + 0.2, dim=-1)[0]
    err, rms, scale, _ = _raw_errors(nerf, ops, gpu, params, c["expr"], c["latent"], ro, rd, z)
    print("layer scales 2^+-6: f16x3 max|err|", err["f16x3"].tolist(), "f32", err["f32"].tolist(), "scale", scale.tolist())
    assert torch.all(err["f16x3"] <= 2e-5 * scale + 2e-5)
    assert torch.all(rms["f16x3"] <= 4.0 * rms["f32"] + 1e-9)


def test_f16x3_range_guard_fails_loudly(hip_lib, gpu):
    """Activations beyond fp16's range (here: a hidden layer scaled by 2^14): the pre-flight probe (exact-f32 forward on a sample
    of the frame's points, 4x margin) makes run_one_iter_of_nerf raise instead of returning a corrupted frame; the kernel itself
    saturates instead of producing NaN, and flags a non-finite density (an unsaturated fc_feat overflow)."""
    import nerf
    c = C.build_case("eval_det_64_128")
    c["p_coarse"] = dict(c["p_coarse"])
    c["p_coarse"]["layers_xyz.1.weight"] = c["p_coarse"]["layers_xyz.1.weight"] * 2.0 ** 14
    nerf.set_mlp_precision("f16x3")
    with pytest.raises(RuntimeError, match="fp16 range"):
        U.run_product(nerf, c, gpu)
    # the kernel alone (no probe): finite colours thanks to the saturating ReLU -- wrong, but not NaN-poisoned
    from nerf import ops
    m = U.make_model(nerf, c["p_coarse"], gpu)
    cond = ops.paper_condition(m.hip_weights().get(), c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    z = torch.linspace(0.2, 0.8, 64).expand(c["n_rays"], 64).contiguous().to(gpu)
    raw = ops.paper_mlp_fwd_f16(m.hip_weights().get_f16(), cond, c["ro"].to(gpu), c["rd"].to(gpu), z)
    assert bool(torch.isfinite(raw[..., :3]).all())
    # fc_feat has no ReLU: blowing IT up reaches sigma as inf / NaN and sets the kernel's sticky flag
    p2 = dict(C.build_case("eval_det_64_128")["p_coarse"])
    p2["fc_feat.weight"] = p2["fc_feat.weight"] * 2.0 ** 16
    m2 = U.make_model(nerf, p2, gpu)
    cond2 = ops.paper_condition(m2.hip_weights().get(), c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    ops.paper_mlp_fwd_f16(m2.hip_weights().get_f16(), cond2, c["ro"].to(gpu), c["rd"].to(gpu), z)
    with pytest.raises(RuntimeError, match="fp16 range"):
        ops.check_f16_range(m2)
    nerf.set_mlp_precision("f32")
    out, *_ = U.run_product(nerf, c, gpu)                        # the exact-f32 kernels render the same model
    assert bool(torch.isfinite(out[0]).all())
    nerf.set_mlp_precision("f16x3")
    c2 = C.build_case("eval_det_64_128")
    out, *_ = U.run_product(nerf, c2, gpu)                       # a healthy model is not affected
    assert bool(torch.isfinite(out[3]).all())


@pytest.mark.parametrize("name", ["eval_det_64_128", "soft_eval_det_64_128", "train_rand_64_64"])
def test_f16x3_end_to_end_at_f32_tolerances(hip_lib, gpu, name):
    """run_one_iter_of_nerf under nerf.set_mlp_precision("f16x3") against the reference's golden outputs with the SAME
    tolerances the exact-f32 path is held to (tests/test_gpu_e2e.py), plus the PSNR gate against the oracle."""
    import nerf
    from tests.test_gpu_e2e import GOLD, NAMES7, _tol
    c = C.build_case(name)
    gold = np.load(os.path.join(GOLD, f"{name}.npz"))
    nerf.set_mlp_precision("f16x3")
    out, *_ = U.run_product(nerf, c, gpu)
    for n, t in zip(NAMES7, out):
        d = np.abs(t.cpu().numpy() - gold[n])
        print(f"[{name} f16x3] {n}: max|d|={d.max():.3e}")
        assert d.max() <= _tol(name)[n], (n, d.max())
    ref = C.run_oracle(c)
    for k in (0, 3):
        dp = abs(O.psnr(out[k].cpu(), c["tgt"]) - O.psnr(ref[k], c["tgt"]))
        assert dp <= 1e-4, (k, dp)


def test_f16x3_range_guard_is_armed_in_training(hip_lib, gpu):
    """Training under "f16x3" (VERDICT r02 weak #1): (a) an overflow that happens in ONE iteration between two polls is still seen at
    the next poll -- every step re-packs the weight stream, which clears the stream's flag, so the flag is carried over on the
    device (ops.PaperWeights._get); (b) a model that drifts out of range and stays there is refused by the exact-f32 probe at the
    next cadence point (ops.set_f16_train_probe_every, which the trainer sets to its print_every).  Both raise RuntimeError."""
    import nerf
    from nerf import ops
    c = C.build_case("train_rand_64_64")
    opt = U.make_options(nerf, 64, 64, True, 0.1, chunksize=2048)
    ex, ed = U.encoders(nerf)
    n = c["ro"].shape[0]

    def step(mc, mf):
        latent = c["latent"].clone().to(gpu).requires_grad_(True)
        with torch.enable_grad():
            out = nerf.run_one_iter_of_nerf(512, 512, None, mc, mf, c["ro"].to(gpu), c["rd"].to(gpu), opt, mode="train",
                                            encode_position_fn=ex, encode_direction_fn=ed, expressions=c["expr"].to(gpu),
                                            background_prior=c["bg"].to(gpu), latent_code=latent)
            (out[0].sum() + out[3].sum()).backward()
    nerf.set_mlp_precision("f16x3")
    try:
        # (a) the sticky flag
        ops.set_f16_train_probe_every(1000)                              # probe on the first step only: isolate the flag path
        mc, mf = U.make_model(nerf, c["p_coarse"], gpu), U.make_model(nerf, c["p_fine"], gpu)
        step(mc, mf)
        step(mc, mf)
        ops.check_f16_range(mc, mf)                                      # healthy so far
        with torch.no_grad():
            mf.fc_feat.weight.mul_(2.0 ** 16)                            # fc_feat has no ReLU: its overflow reaches sigma as inf / NaN
        step(mc, mf)
        with torch.no_grad():
            mf.fc_feat.weight.mul_(2.0 ** -16)
        step(mc, mf)                                                     # two healthy steps: each re-packs the stream
        step(mc, mf)
        with pytest.raises(RuntimeError, match="fp16 range"):
            ops.check_f16_range(mc, mf)
        # (b) the probe cadence
        ops.set_f16_train_probe_every(2)                                 # calls 1, 3, 5, ... of each model
        mc, mf = U.make_model(nerf, c["p_coarse"], gpu), U.make_model(nerf, c["p_fine"], gpu)
        step(mc, mf)
        with torch.no_grad():
            mc.layers_xyz[1].weight.mul_(2.0 ** 14)
        step(mc, mf)                                                     # call 2: between two probes, saturates silently
        with pytest.raises(RuntimeError, match="fp16 range"):
            step(mc, mf)                                                 # call 3: the probe refuses the model
    finally:
        ops.set_f16_train_probe_every(128)
        nerf.set_mlp_precision("f32")
