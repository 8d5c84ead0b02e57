"""Gradient parity of the HIP training path against autograd on the CPU oracle (and the reference's own autograd
via the golden fixture).  GPU only.  Gate: relative L2 error <= 1e-4 per parameter tensor."""
import os

import numpy as np
import pytest
import torch

from oracle import cases as C
from oracle import nerface_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("n_rays,s,bgflag,noisy", [(6, 64, True, True), (5, 192, True, False), (4, 7, False, False), (3, 130, True, True)])
def test_volume_render_bwd(hip_lib, gpu, n_rays, s, bgflag, noisy):
    from nerf import ops
    g = torch.Generator().manual_seed(21)
    raw = (torch.randn((n_rays, s, 4), generator=g) * 2.0).double()
    raw[..., 3] = raw[..., 3] * 8
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0].double()
    rd = torch.randn((n_rays, 3), generator=g).double()
    bg = torch.rand((n_rays, 3), generator=g).double() if bgflag else None
    noise = (torch.randn((n_rays, s), generator=g) * 0.1).double() if noisy else None
    d_rgb = torch.randn((n_rays, 3), generator=g).double()
    raw_l = raw.clone().requires_grad_(True)
    raw_in = raw_l
    if bg is not None:                                  # T:95-96: the overwrite kills the gradient of the last colour
        raw_in = torch.cat((raw_l[:, :-1], torch.cat((bg[:, None, :], raw_l[:, -1:, 3:]), dim=-1)), dim=1)
    rgb, *_ = O.volume_render(raw_in, z, rd, noise, has_background=bgflag)
    rgb.backward(d_rgb)
    f = lambda t: None if t is None else t.float().to(gpu).contiguous()
    d_raw = ops.volume_render_bwd(f(raw), f(z), f(rd), f(noise), f(bg), f(d_rgb)).cpu()
    err = rel_l2(d_raw, raw_l.grad)
    print("volume_render_bwd rel L2", err)
    assert err < 2e-5


# sections of the saved-activation buffer (csrc/nf_mlp_layout.h: nfl::S_*): (offset, width)
SAVED = dict(pe=(0, 64), h0=(64, 256), h1=(320, 256), h2=(576, 256), h3=(832, 256), h4=(1088, 256), h5=(1344, 256),
             feat=(1600, 256), d0=(1856, 128), d1=(1984, 128), d2=(2112, 128), dirf=(2240, 16))
RELU_ORDER = ["h0", "h1", "h2", "h3", "h4", "h5", "d0", "d1", "d2"]


def saved_section(saved, name, n_points):
    off, w = SAVED[name]
    return saved[off * n_points:(off + w) * n_points].view(n_points, w)


def relu_masks(sv, n_points, mode):
    """The nine ReLU decisions (RELU_ORDER) a training forward saved, as boolean (n_points, width) matrices.  Exact f32: the saved
    post-ReLU activations are the f32 values themselves, x > 0.  Split forwards: the BIT MASKS their backward chain reads
    (S_MASK: [layer][point][lane half h][4 dwords], bit 16 (nt & 1) + r of dword nt >> 1 <-> feature 32 nt + (r & 3) + 8 (r >> 2) + 4 h)
    -- not the sign of the (hi, lo) pair, which is 0 for a positive activation below the pair's underflow (fp16: 1.9e-9)."""
    if mode in ("f32", False, None):
        return [saved_section(sv, k, n_points) > 0 for k in RELU_ORDER]
    words = sv[2256 * n_points:(2256 + 72) * n_points].view(torch.int32).view(9, n_points, 2, 4)
    out = []
    for l, k in enumerate(RELU_ORDER):
        width = SAVED[k][1]
        m = torch.zeros((n_points, width), dtype=torch.bool, device=sv.device)
        for nt in range(width // 32):
            for h in range(2):
                w = words[l, :, h, nt >> 1]
                for r in range(16):
                    m[:, 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * h] = ((w >> (16 * (nt & 1) + r)) & 1).bool()
        out.append(m)
    return out


def f32_rows(saved_t, n_points, mode):
    """`saved` of a training forward in the exact-f32 layout the section table above describes.  The split forwards write their
    hidden-layer outputs as the weight-gradient kernel's (hi, lo) fragment stream (csrc/nf_mlp_bf16_machinery.inc); the library's own
    converter (nf_split_saved_to_f32: x = hi + lo) turns them back into f32 rows."""
    from nerf import ops
    return saved_t if mode in ("f32", False, None) else ops.split_saved_to_f32(saved_t, n_points, f16=mode in ("f16", "f16x3"))


def _oracle_mlp_grads(p, ro, rd, z, expr, latent, d_raw, masks=None, dtype=torch.float64):
    pp = {k: v.to(dtype).clone().requires_grad_(True) for k, v in p.items()}
    lat = latent.to(dtype).clone().requires_grad_(True)
    x = O.encode_points(ro.to(dtype), rd.to(dtype), z.to(dtype), O.NEAR, O.FAR)
    acts = []
    out = O.paper_mlp(pp, x, expr.to(dtype), lat, masks=masks, acts=acts)
    out.backward(d_raw.reshape(-1, 4).to(dtype))
    return pp, lat, acts


@pytest.mark.parametrize("n_rays,s", [(8, 64), (3, 7), (37, 128)])
def test_bf16x3_training_forward_saves_match_f32(hip_lib, gpu, n_rays, s):
    """The split-bf16 training forward must fill the `saved` buffer (all sections, same layout) like the exact-f32 one."""
    import nerf
    from nerf import ops
    c = C.build_case("train_rand_64_64")
    g = torch.Generator().manual_seed(17)
    ro, rd, _, _, _ = C.ray_subset(512, 512, 9, n_rays, 17)
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0]
    m = U.make_model(nerf, c["p_fine"], gpu)
    hw = m.hip_weights()
    cond = ops.paper_condition(hw.get(), c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    raw_f, (sv_f,) = ops.paper_mlp_fwd_train(hw.get(), cond, ro.to(gpu), rd.to(gpu), z.to(gpu))
    raw_b, (sv_b,) = ops.paper_mlp_fwd_train(hw.get(), cond, ro.to(gpu), rd.to(gpu), z.to(gpu), packed_b=hw.get_bf16())
    n_pts = n_rays * s
    sv_b = f32_rows(sv_b, n_pts, "bf16")
    for name in SAVED:
        a, b = saved_section(sv_f.cpu(), name, n_pts), saved_section(sv_b.cpu(), name, n_pts)
        d = float((a - b).abs().max())
        assert d <= 3e-4 * (1 + float(a.abs().max())), (name, d)
    assert torch.equal(saved_section(sv_f.cpu(), "dirf", n_pts), saved_section(sv_b.cpu(), "dirf", n_pts))
    assert float((raw_f - raw_b).abs().max()) < 3e-4 * (1 + float(raw_f.abs().max()))


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("n_rays,s", [(8, 64), (3, 7), (37, 128)])
def test_paper_mlp_bwd(hip_lib, gpu, n_rays, s, split):
    import nerf
    from nerf import ops
    c = C.build_case("train_rand_64_64")
    g = torch.Generator().manual_seed(13)
    ro, rd, _, _, _ = C.ray_subset(512, 512, 9, n_rays, 13)
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0]
    d_raw = torch.randn((n_rays, s, 4), generator=g)
    p = c["p_fine"]
    m = U.make_model(nerf, p, gpu)
    pk = m.hip_weights().get()
    cond = ops.paper_condition(pk, c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    raw_t, saved = ops.paper_mlp_fwd_train(pk, cond, ro.to(gpu), rd.to(gpu), z.to(gpu), packed_b=m.hip_weights().get_bf16() if split else None)
    raw_e = (ops.paper_mlp_fwd_bf16(m.hip_weights().get_bf16(), cond, ro.to(gpu), rd.to(gpu), z.to(gpu)) if split
             else ops.paper_mlp_fwd(pk, cond, ro.to(gpu), rd.to(gpu), z.to(gpu)))
    assert torch.equal(raw_t, raw_e)                       # training forward == eval forward, bit for bit
    # saved activations against the oracle (spot check: fc_feat output and PE slots that hold raw xyz)
    grads, g_lat = ops.paper_mlp_bwd(m, pk, cond, z.to(gpu), d_raw.to(gpu), saved, split=split)
    # ReLU masks as the HIP forward saw them: the oracle's backward is evaluated with the same masks so that the
    # comparison measures the backward arithmetic, not the handful of units whose pre-activation rounds across 0
    n_pts = n_rays * s
    sv = f32_rows(saved[0], n_pts, "bf16" if split else "f32").cpu()
    masks = relu_masks(sv, n_pts, "bf16" if split else "f32")
    pp, lat, acts = _oracle_mlp_grads(p, ro, rd, z, c["expr"], c["latent"], d_raw, masks=masks)
    _, _, acts_free = _oracle_mlp_grads(p, ro, rd, z, c["expr"], c["latent"], d_raw, masks=None)
    order = RELU_ORDER[:6] + ["feat"] + RELU_ORDER[6:]
    flips = 0
    for name, a_free in zip(order, acts_free):
        got = saved_section(sv, name, n_pts)
        assert (got.double() - a_free).abs().max() < 3e-4 * (1 + float(a_free.detach().abs().max())), name     # saved activations
        if name != "feat":
            flips += int(((got > 0) != (a_free > 0)).sum())
    print(f"ReLU mask flips vs fp64 oracle: {flips} of {sum(mk.numel() for mk in masks)}")
    assert flips <= 1e-5 * sum(mk.numel() for mk in masks) + 2
    worst = 0.0
    for k, gh in zip(ops.PAPER_KEYS, grads):
        go = pp[k].grad
        if k.startswith("layers_dir.3"):
            assert gh is None and (go is None or float(go.abs().max()) == 0.0)
            continue
        e = rel_l2(gh.cpu(), go)
        worst = max(worst, e)
        assert e < 1e-4, (k, e)      # SURVEY §8(d)(iii) gate in BOTH arithmetics (split-bf16 measured worst: 6e-5)
    e = rel_l2(g_lat.cpu(), lat.grad)
    print(f"mlp bwd ({n_rays}x{s}, split={split}): worst param rel L2 {worst:.2e}, latent {e:.2e}")
    assert e < 1e-4
    if split:     # the split-bf16 dW GEMMs against the exact-f32 ones on the SAME saved activations and dZ
        grads_x, g_lat_x = ops.paper_mlp_bwd(m, pk, cond, z.to(gpu), d_raw.to(gpu), saved, split=True, exact_dw=True)
        for (k, _), gs, gx in zip(m.named_parameters(), grads, grads_x):
            if gx is None:
                assert gs is None
                continue
            assert rel_l2(gs.cpu(), gx.cpu()) < 1e-4, (k, rel_l2(gs.cpu(), gx.cpu()))
        assert rel_l2(g_lat.cpu(), g_lat_x.cpu()) < 1e-4


@pytest.mark.parametrize("n_rays,s", [(8, 64), (3, 7), (37, 128)])
def test_paper_mlp_bwd_f16x3(hip_lib, gpu, n_rays, s):
    """The three training GEMM kernels on fp16 operand pairs: gradients against the fp64 oracle at the masks the HIP forward saw,
    held to the EXACT-f32 kernels' gate (1e-4 per tensor) and compared with the f32 and split-bf16 kernels' own errors."""
    import nerf
    from nerf import ops
    c = C.build_case("train_rand_64_64")
    g = torch.Generator().manual_seed(13)
    ro, rd, _, _, _ = C.ray_subset(512, 512, 9, n_rays, 13)
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0]
    d_raw = torch.randn((n_rays, s, 4), generator=g) * 1e-4          # realistic scale: d loss / d raw of a 2048-ray batch
    p = c["p_fine"]
    m = U.make_model(nerf, p, gpu)
    hw = m.hip_weights()
    pk = hw.get()
    cond = ops.paper_condition(pk, c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    dv = lambda t: t.to(gpu)
    worst = {}
    for mode in ("f32", "f16", "bf16"):
        raw_t, saved = ops.paper_mlp_fwd_train(pk, cond, dv(ro), dv(rd), dv(z), packed_b=hw.get_bf16() if mode == "bf16" else None,
                                               packed_h=hw.get_f16() if mode == "f16" else None)
        if mode == "f16":
            assert torch.equal(raw_t, ops.paper_mlp_fwd_f16(hw.get_f16(), cond, dv(ro), dv(rd), dv(z)))      # training forward == eval forward
        grads, g_lat = ops.paper_mlp_bwd(m, pk, cond, dv(z), dv(d_raw), saved, split={"f32": False, "f16": "f16", "bf16": True}[mode])
        n_pts = n_rays * s
        sv = f32_rows(saved[0], n_pts, mode).cpu()
        masks = relu_masks(sv, n_pts, mode)
        pp, lat, _ = _oracle_mlp_grads(p, ro, rd, z, c["expr"], c["latent"], d_raw, masks=masks)
        w = 0.0
        for k, gh in zip(ops.PAPER_KEYS, grads):
            if k.startswith("layers_dir.3"):
                assert gh is None
                continue
            w = max(w, rel_l2(gh.cpu(), pp[k].grad))
        worst[mode] = (w, rel_l2(g_lat.cpu(), lat.grad))
    print(f"mlp bwd ({n_rays}x{s}) worst param rel L2 / latent:  f32 {worst['f32'][0]:.2e} / {worst['f32'][1]:.2e}   "
          f"f16x3 {worst['f16'][0]:.2e} / {worst['f16'][1]:.2e}   bf16x3 {worst['bf16'][0]:.2e} / {worst['bf16'][1]:.2e}")
    assert worst["f16"][0] < 1e-4 and worst["f16"][1] < 1e-4
    assert worst["f16"][0] < max(4.0 * worst["f32"][0], 5e-6)           # fp32-class: within a small factor of the exact-f32 kernels


@pytest.mark.parametrize("case", ["soft_train_noflip_64_64", "train_rand_64_64"])
def test_train_step_f16x3_follows_the_f32_path(hip_lib, gpu, case):
    """Whole training step (coarse + fine, noise, perturb, latent regulariser) under nerf.set_mlp_precision("f16x3") against the
    same step on the exact-f32 kernels: loss and every gradient tensor.  On the no-flip frame (oracle/cases.py) every tensor agrees to
    1e-4; on the hard family (x1000 density head) the odd ReLU unit whose input lies within rounding of zero takes different sides in
    the two arithmetics and moves the few tensors upstream of it by ~1e-3 each (which units do changes with every bit of the inputs:
    round 5's branch-free sincos moved the worst tensor from 6e-4 to 4.9e-3), so there the gate is on the MEDIAN tensor -- an error
    of the arithmetic would show in all of them -- with a loose bound on the worst."""
    import nerf
    c = C.build_case(case)
    res = {}
    for prec in ("f32", "f16x3"):
        nerf.set_mlp_precision(prec)
        out, mc, mf, latent = U.run_product(nerf, c, gpu, mode="train", grad=True)
        loss = O.train_loss(out[0], out[3], c["tgt"].to(gpu), latent)
        loss.backward()
        res[prec] = (float(loss), {f"{t}.{k}": v.grad.clone() for t, m in (("c", mc), ("f", mf)) for k, v in m.named_parameters() if v.grad is not None},
                     latent.grad.clone())
    nerf.set_mlp_precision("f32")
    assert abs(res["f32"][0] - res["f16x3"][0]) < 2e-6
    errs = sorted(rel_l2(res["f16x3"][1][k], v) for k, v in res["f32"][1].items())
    worst, median, e_lat = errs[-1], errs[len(errs) // 2], rel_l2(res["f16x3"][2], res["f32"][2])
    print(f"train step f16x3 vs f32 kernels [{case}]: worst param rel L2 {worst:.2e}, median {median:.2e}, latent {e_lat:.2e}")
    if case == "soft_train_noflip_64_64":
        assert worst < 1e-4 and e_lat < 1e-4
    else:
        n_loose = sum(e >= 1.5e-3 for e in errs)
        print(f"    {n_loose} of {len(errs)} tensors above 1.5e-3 (allowed: {MAX_LOOSE_TENSORS})")
        assert median < 3e-4 and worst < 2e-2 and e_lat < 1e-4 and n_loose <= MAX_LOOSE_TENSORS


MAX_LOOSE_TENSORS = 4            # of 48 (two networks x 24 live tensors); measured on MI355X: 2 (the layer of the flipped unit and the one below it)


def test_train_step_gradients_vs_oracle_and_reference(hip_lib, gpu):
    """Full training step (coarse+fine, noise, perturb, latent regulariser) through run_one_iter_of_nerf + autograd."""
    import nerf
    c = C.build_case("train_rand_64_64")
    out, mc, mf, latent = U.run_product(nerf, c, gpu, mode="train", grad=True)
    loss = O.train_loss(out[0], out[3], c["tgt"].to(gpu), latent)
    loss.backward()
    gold = np.load(os.path.join(GOLD, "train_rand_64_64_grads.npz"))
    assert abs(float(loss) - float(gold["loss"])) < 2e-6
    # oracle autograd in fp64 on identical inputs
    pc = {k: v.double().clone().requires_grad_(True) for k, v in c["p_coarse"].items()}
    pf = {k: v.double().clone().requires_grad_(True) for k, v in c["p_fine"].items()}
    lat = c["latent"].double().clone().requires_grad_(True)
    d = lambda t: None if t is None else t.double()
    o = O.render_rays(pc, pf, d(c["ro"]), d(c["rd"]), d(c["expr"]), lat, d(c["bg"]), O.NEAR, O.FAR, 64, 64, t_rand=d(c["t_rand"]),
                      noise_c=d(c["noise_c"]), u=d(c["u"]), noise_f=d(c["noise_f"]))
    O.train_loss(o[0], o[3], d(c["tgt"]), lat).backward()
    worst, errs, nerrs = 0.0, [], []
    for tag, m, po in (("coarse", mc, pc), ("fine", mf, pf)):
        for k, v in m.named_parameters():
            if k.startswith("layers_dir.3"):
                assert v.grad is None                                     # Q3, and `none:` entries of the fixture
                assert f"none:{tag}.{k}" in gold.files
                continue
            e = rel_l2(v.grad.cpu(), po[k].grad)
            worst = max(worst, e)
            errs.append(e)
            # End to end, fp32 vs fp64 differ by the odd ReLU unit (MLP or the density ReLU of V:52) whose
            # pre-activation rounds across zero -- each flip moves a gradient tensor by O(1/sqrt(#points)) -- and the fine
            # pass inherits the resampled-depth sensitivity (test_gpu_e2e.TOL).  The 1e-4 gate on the backward arithmetic
            # itself is enforced mask-consistently in test_paper_mlp_bwd / test_volume_render_bwd above.
            # Which units flip changes with every bit of the inputs (round 4: worst tensor 6.2e-4, one flipped unit of 24 rays x 128 points),
            # so the per-tensor bound is loose and the MEDIAN tensor carries the gate below: a defect of the arithmetic shows in all of them.
            # The same step WITHOUT a decision in reach is pinned at 1e-4 per tensor, full tensors against the reference's autograd, in
            # test_train_step_gradients_full_tensors_vs_reference_autograd.
            assert e < 2e-2, (tag, k, e)
            want = float(gold[f"norm:{tag}.{k}"])
            nerrs.append(abs(float(v.grad.double().norm()) - want) / (want + 1e-30))
            assert nerrs[-1] <= 1e-2, (tag, k)
    errs.sort(), nerrs.sort()
    assert errs[len(errs) // 2] < 2e-4 and nerrs[len(nerrs) // 2] < 1e-4, (errs[len(errs) // 2], nerrs[len(nerrs) // 2])
    # the loose per-tensor bound alone would let a defect confined to a few tensors through -- a flipped unit moves the
    # tensors of ITS layer and the ones downstream of it, so only a handful may sit above the round-4 bound (1.5e-3)
    n_loose = sum(e >= 1.5e-3 for e in errs)
    print(f"train step: {n_loose} of {len(errs)} tensors above 1.5e-3 (allowed: {MAX_LOOSE_TENSORS})")
    assert n_loose <= MAX_LOOSE_TENSORS, errs[-8:]
    e_lat = rel_l2(latent.grad.cpu(), lat.grad)
    e_ref = rel_l2(latent.grad.cpu(), torch.from_numpy(gold["latent"]))
    print(f"train step: worst param rel L2 {worst:.2e}, median {errs[len(errs) // 2]:.2e}; latent vs oracle(fp64) {e_lat:.2e}, vs reference autograd {e_ref:.2e}")
    assert e_lat < 1e-4 and e_ref < 1e-4                     # measured 7.6e-6 / 7.1e-6


# gate of the no-flip case: SURVEY §8(d)(iii)'s own 1e-4 per tensor in EVERY arithmetic (measured on MI355X: f32 1.7e-6, f16x3 2.0e-6,
# bf16x3 7.3e-6 worst tensor; latent 8e-7 / 1.3e-6 / 3.3e-6 -- no unit flips on this frame even at bf16x3's 16 significand bits)
NOFLIP_GATE = {"f32": 1e-4, "f16x3": 1e-4, "bf16x3": 1e-4}


@pytest.mark.parametrize("precision", ["f32", "f16x3", "bf16x3"])
def test_train_step_gradients_full_tensors_vs_reference_autograd(hip_lib, gpu, precision):
    """End-to-end gradient at SURVEY §8(d)(iii)'s gate: one whole training step (run_one_iter_of_nerf in train mode with jitter, density
    noise, resampling, both networks, the latent regulariser; loss.backward()) against the REFERENCE's own autograd, every parameter tensor
    and the latent row in full (tests/golden/soft_train_noflip_64_64_grads.npz; TR:355-392).  The case is the soft-family frame on which no
    ReLU decision sits within fp32 rounding of zero (oracle/make_golden.py `soft_grads search`), so rel-L2 measures the backward arithmetic."""
    import nerf
    c = C.build_case("soft_train_noflip_64_64")
    gold = np.load(os.path.join(GOLD, "soft_train_noflip_64_64_grads.npz"))
    nerf.set_mlp_precision(precision)
    try:
        out, mc, mf, latent = U.run_product(nerf, c, gpu, mode="train", grad=True)
        loss = O.train_loss(out[0], out[3], c["tgt"].to(gpu), latent)
        loss.backward()
    finally:
        nerf.set_mlp_precision("f32")
    assert abs(float(loss) - float(gold["loss"])) < (2e-6 if precision != "bf16x3" else 2e-5)
    worst, gate = ("", 0.0), NOFLIP_GATE[precision]
    for tag, m in (("coarse", mc), ("fine", mf)):
        for k, v in m.named_parameters():
            if f"none:{tag}.{k}" in gold.files:
                assert v.grad is None                                      # Q3
                continue
            e = rel_l2(v.grad.cpu(), torch.from_numpy(gold[f"full:{tag}.{k}"]))
            worst = max(worst, (f"{tag}.{k}", e), key=lambda t: t[1])
            assert e < gate, (precision, tag, k, e)
    e_lat = rel_l2(latent.grad.cpu(), torch.from_numpy(gold["latent"]))
    print(f"train step {precision}, full tensors vs reference autograd: worst {worst[0]} {worst[1]:.2e}, latent {e_lat:.2e} (gate {gate:g})")
    assert e_lat < gate


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
def test_fused_optimizer_updates_reach_the_kernels(hip_lib, gpu, precision):
    """torch's FUSED optimizers update parameters without bumping their version counters, which the packed weight images were
    once keyed on alone: the kernels kept evaluating the initial weights.  Every run_one_iter_of_nerf call now advances the pack
    epoch, so the step after Adam(fused=True).step() sees the new weights: same rays, same random draws, different output --
    and the same output as a model that was given the updated weights from scratch."""
    import nerf
    c = C.build_case("train_rand_64_64")
    nerf.set_mlp_precision(precision)
    mc, mf = U.make_model(nerf, c["p_coarse"], gpu), U.make_model(nerf, c["p_fine"], gpu)
    table = torch.zeros((3, 32), device=gpu, requires_grad=True)
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()) + [table], lr=1e-3, fused=True)
    o = U.make_options(nerf, 64, 64, True, 0.1)
    ex, ed = U.encoders(nerf)
    rands, randns = U.case_random_lists(c)

    def render(a, b, grad):
        ctx = torch.enable_grad() if grad else torch.no_grad()
        with ctx, U.injected_random(rands, randns):
            return nerf.run_one_iter_of_nerf(512, 512, None, a, b, c["ro"].to(gpu), c["rd"].to(gpu), o, mode="train", encode_position_fn=ex,
                                             encode_direction_fn=ed, expressions=c["expr"].to(gpu), background_prior=c["bg"].to(gpu),
                                             latent_code=table[1])
    out0 = render(mc, mf, True)
    O.train_loss(out0[0], out0[3], c["tgt"].to(gpu), table[1]).backward()
    versions = [p._version for p in mc.parameters()]
    opt.step()
    out1 = render(mc, mf, False)
    assert float((out1[3] - out0[3].detach()).abs().max()) > 1e-5                     # the update is visible ...
    mc2, mf2 = U.make_model(nerf, mc.state_dict(), gpu), U.make_model(nerf, mf.state_dict(), gpu)
    out2 = render(mc2, mf2, False)
    assert torch.equal(out1[3], out2[3]) and torch.equal(out1[0], out2[0])            # ... and it is exactly the updated model
    print("fused Adam bumped version counters:", versions != [p._version for p in mc.parameters()])


def test_adam_step_moves_live_parameters(hip_lib, gpu):
    """The trainer's loop (TR:389-392): backward, Adam over [coarse, fine, latent table], zero_grad; the cached weight
    image must follow the in-place update (version counters) and the gradient must land in the latent ROW."""
    import nerf
    c = C.build_case("train_rand_64_64")
    mc, mf = U.make_model(nerf, c["p_coarse"], gpu), U.make_model(nerf, c["p_fine"], gpu)
    table = torch.zeros((5, 32), device=gpu, requires_grad=True)
    opt = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()) + [table], lr=1e-3)
    o = U.make_options(nerf, 64, 64, True, 0.1)
    ex, ed = U.encoders(nerf)
    losses = []
    for it in range(3):
        torch.manual_seed(100 + it)
        out = nerf.run_one_iter_of_nerf(512, 512, None, mc, mf, c["ro"].to(gpu), c["rd"].to(gpu), o, mode="train",
                                        encode_position_fn=ex, encode_direction_fn=ed, expressions=c["expr"].to(gpu),
                                        background_prior=c["bg"].to(gpu), latent_code=table[2])
        loss = O.train_loss(out[0], out[3], c["tgt"].to(gpu), table[2])
        loss.backward()
        if it == 0:
            assert float(table.grad[2].abs().sum()) > 0 and float(table.grad[[0, 1, 3, 4]].abs().sum()) == 0.0
            assert mc.layers_dir[3].weight.grad is None
        opt.step()
        opt.zero_grad()
        losses.append(float(loss))
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0]                    # 3 Adam steps on the same rays/target reduce the loss


@pytest.mark.parametrize("precision", ["f32", "f16x3", "bf16x3"])
def test_training_trajectory_follows_oracle(hip_lib, gpu, precision):
    """Eight Adam steps (coarse+fine, perturb + density noise, latent regulariser) on the HIP path and on the CPU oracle with
    identical data and random draws: the loss trajectories must coincide.  (Adam normalises tiny gradients, so parameters
    are compared through the loss they produce, not element-wise.)"""
    import nerf
    c = C.build_case("train_rand_64_64")
    n, nc, nf = c["n_rays"], 64, 64
    # --- oracle side (fp32 autograd on CPU)
    pc = {k: v.clone().requires_grad_(True) for k, v in c["p_coarse"].items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in c["p_fine"].items()}
    lat_o = torch.zeros(32, requires_grad=True)
    opt_o = torch.optim.Adam(list(pc.values()) + list(pf.values()) + [lat_o], lr=5e-4)
    # --- product side
    mc, mf = U.make_model(nerf, c["p_coarse"], gpu), U.make_model(nerf, c["p_fine"], gpu)
    lat_h = torch.zeros(32, device=gpu, requires_grad=True)
    opt_h = torch.optim.Adam(list(mc.parameters()) + list(mf.parameters()) + [lat_h], lr=5e-4)
    o = U.make_options(nerf, nc, nf, True, 0.1)
    ex, ed = U.encoders(nerf)
    nerf.set_mlp_precision(precision)
    losses_o, losses_h = [], []
    try:
        for it in range(8):
            t_rand, noise_c, u, noise_f = C.randoms(n, nc, nf, seed=500 + it)
            out = O.render_rays(pc, pf, c["ro"], c["rd"], c["expr"], lat_o, c["bg"], O.NEAR, O.FAR, nc, nf, t_rand=t_rand,
                                noise_c=noise_c * 0.1, u=u, noise_f=noise_f * 0.1)
            lo = O.train_loss(out[0], out[3], c["tgt"], lat_o)
            opt_o.zero_grad()
            lo.backward()
            opt_o.step()
            losses_o.append(float(lo.detach()))
            with U.injected_random([t_rand, u], [noise_c, noise_f]):
                outh = nerf.run_one_iter_of_nerf(512, 512, None, mc, mf, c["ro"].to(gpu), c["rd"].to(gpu), o, mode="train",
                                                 encode_position_fn=ex, encode_direction_fn=ed, expressions=c["expr"].to(gpu),
                                                 background_prior=c["bg"].to(gpu), latent_code=lat_h)
            lh = O.train_loss(outh[0], outh[3], c["tgt"].to(gpu), lat_h)
            opt_h.zero_grad()
            lh.backward()
            opt_h.step()
            losses_h.append(float(lh.detach()))
    finally:
        nerf.set_mlp_precision("f32")
    print(precision, "oracle losses", [f"{v:.6f}" for v in losses_o])
    print(precision, "HIP    losses", [f"{v:.6f}" for v in losses_h])
    assert losses_o[-1] < losses_o[0]
    for a, b in zip(losses_o, losses_h):
        assert abs(a - b) <= 2e-4 * abs(a) + 1e-6, (losses_o, losses_h)


# per-tensor gates of test_training_kernels_vs_fp64_at_training_size.  The bias gradients are sums over 262144 points in which the
# terms cancel to 1/300 ... 1/12000 of their magnitude (fc_feat.bias), so per-element error is amplified that much: the exact-f32
# kernels land at 3e-5 there, the fp16 ones below that (measured 1.4e-5), the split-bf16 ones (2^-16 per element) at 1e-4.
TRAIN_SIZE_GATE = {"f32": 1e-4, "f16": 1e-4, "bf16": 4e-4}


@pytest.mark.parametrize("spread", [0, 6])
@pytest.mark.parametrize("n_rays,s", [(2048, 128), (2047, 127)])
def test_training_kernels_vs_fp64_at_training_size(hip_lib, gpu, n_rays, s, spread):
    """BASELINE training size (2048 rays x 128 samples, plus a ragged 2047 x 127): each training arithmetic (exact f32, split
    fp16, split bf16) -- its own training forward, backward chain and weight-gradient GEMMs -- against fp64 autograd of the
    oracle MLP evaluated ON THE DEVICE at the ReLU masks that forward saved (mask-consistent: at this size two forwards differ
    in a few of the 6e8 ReLU decisions, which moves gradients by 1e-3 and says nothing about the kernels).  Upstream gradient
    at the realistic 1 / (3 n_rays) scale; spread = 6 multiplies it point by point by 10^-U(0, 6), the dynamic range a
    volume-rendering loss has, which is what the per-point gradient scaling of the fp16 chain is for.  Besides the gates:
    the fp16 kernels must be at least as accurate as the exact-f32 ones (factor 2 slack), tensor class by tensor class."""
    import nerf
    from nerf import ops
    c = C.build_case("train_rand_64_64")
    g = torch.Generator().manual_seed(29)
    ro, rd, _, _, _ = C.ray_subset(512, 512, 9, n_rays, 29)
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0]
    d_raw = torch.randn((n_rays, s, 4), generator=g) * (1.0 / (n_rays * 3))
    if spread:
        d_raw = d_raw * torch.pow(10.0, -spread * torch.rand((n_rays, s, 1), generator=g))
    p = c["p_fine"]
    m = U.make_model(nerf, p, gpu)
    hw = m.hip_weights()
    pk = hw.get()
    cond = ops.paper_condition(pk, c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    dv = lambda t: t.to(gpu)
    n_pts = n_rays * s
    x64 = O.encode_points(dv(ro).double(), dv(rd).double(), dv(z).double(), O.NEAR, O.FAR)
    worst, saved_act = {}, {}
    for mode in ("f32", "f16", "bf16"):
        _, saved = ops.paper_mlp_fwd_train(pk, cond, dv(ro), dv(rd), dv(z), packed_b=hw.get_bf16() if mode == "bf16" else None,
                                           packed_h=hw.get_f16() if mode == "f16" else None)
        grads, g_lat = ops.paper_mlp_bwd(m, pk, cond, dv(z), dv(d_raw), saved, split={"f32": False, "f16": "f16", "bf16": True}[mode])
        assert all(bool(torch.isfinite(x).all()) for x in grads if x is not None) and bool(torch.isfinite(g_lat).all())
        sv = f32_rows(saved[0], n_pts, mode)
        masks = relu_masks(sv, n_pts, mode)
        saved_act[mode] = sv[:2256 * n_pts]
        pp = {k: v.to(gpu).double().clone().requires_grad_(True) for k, v in p.items()}
        lat = c["latent"].to(gpu).double().clone().requires_grad_(True)
        out = O.paper_mlp(pp, x64, c["expr"].to(gpu).double(), lat, masks=masks)
        out.backward(dv(d_raw).reshape(-1, 4).double())
        errs = {k: rel_l2(gh, pp[k].grad) for k, gh in zip(ops.PAPER_KEYS, grads) if gh is not None}
        errs["latent"] = rel_l2(g_lat, lat.grad)
        worst[mode] = {"weight": max(v for k, v in errs.items() if k.endswith("weight")),
                       "bias": max(v for k, v in errs.items() if k.endswith("bias")), "latent": errs["latent"]}
        print(f"{mode:4s} {n_rays}x{s} spread 1e-{spread}: worst weight {worst[mode]['weight']:.2e} bias {worst[mode]['bias']:.2e} "
              f"latent {worst[mode]['latent']:.2e}")
        for k, v in errs.items():
            assert v < TRAIN_SIZE_GATE[mode], (mode, k, v)
        del pp, lat, out, masks, saved, grads
    for cls in ("weight", "bias", "latent"):
        assert worst["f16"][cls] <= 2 * worst["f32"][cls] + 1e-6, (cls, worst)
    a, b = saved_act["f32"], saved_act["f16"]
    assert float((a - b).abs().max()) <= 2e-5 * (1 + float(a.abs().max()))        # saved activations: f32 rounding apart


@pytest.mark.parametrize("n_rays,s", [(2048, 128), (2047, 127), (2048, 64)])
def test_split_dw_gemm_matches_exact_at_training_size(hip_lib, gpu, n_rays, s):
    """BASELINE training sizes (configs[2]: 2048 rays, 64 coarse / 128 fine samples; plus a ragged size whose last
    16-point stage is partial): on the SAME saved activations and dZ, the split-bf16 weight-gradient kernel (one 16-wave
    workgroup per CU) must agree with the exact-f32 one (28 slices) to the split's 2^-16 class, tensor by tensor."""
    import nerf
    from nerf import ops
    c = C.build_case("train_rand_64_64")
    g = torch.Generator().manual_seed(29)
    ro, rd, _, _, _ = C.ray_subset(512, 512, 9, n_rays, 29)
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0]
    d_raw = torch.randn((n_rays, s, 4), generator=g) * (1.0 / (n_rays * 3))      # mse-like scale
    m = U.make_model(nerf, c["p_fine"], gpu)
    hw = m.hip_weights()
    cond = ops.paper_condition(hw.get(), c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    raw, saved = ops.paper_mlp_fwd_train(hw.get(), cond, ro.to(gpu), rd.to(gpu), z.to(gpu), packed_b=hw.get_bf16())
    g_s, lat_s = ops.paper_mlp_bwd(m, hw.get(), cond, z.to(gpu), d_raw.to(gpu), saved, split=True)
    g_x, lat_x = ops.paper_mlp_bwd(m, hw.get(), cond, z.to(gpu), d_raw.to(gpu), saved, split=True, exact_dw=True)
    worst = 0.0
    for (k, _), a, b in zip(m.named_parameters(), g_s, g_x):
        if b is None:
            assert a is None
            continue
        assert bool(torch.isfinite(a).all()), k
        e = rel_l2(a.cpu(), b.cpu())
        worst = max(worst, e)
        assert e < 1e-4, (k, e)
    e = rel_l2(lat_s.cpu(), lat_x.cpu())
    print(f"split vs exact dW at {n_rays}x{s}: worst rel L2 {worst:.2e}, latent {e:.2e}")
    assert e < 1e-4
    # determinism: the slab reduction has a fixed order
    g_s2, lat_s2 = ops.paper_mlp_bwd(m, hw.get(), cond, z.to(gpu), d_raw.to(gpu), saved, split=True)
    assert all(a2 is None or torch.equal(a, a2) for a, a2 in zip(g_s, g_s2)) and torch.equal(lat_s, lat_s2)


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16x3"])
def test_full_size_training_step_properties(hip_lib, gpu, precision):
    """BASELINE configs[2] at its full size (2048 rays of a 512x512 frame, 64+64 samples, noise 0.1, perturb): the backward is
    linear in the upstream gradient, so doubling the loss must double every gradient EXACTLY (power-of-two scaling commutes
    with fp32 rounding, with the hi/lo bf16 split, and with the fp16 kernels' block floating point, whose per-point scales
    halve when the gradients double); repeating the step with the same random draws must reproduce every
    gradient bit for bit (fixed-order slab reduction, no atomics); every live tensor gets a finite, non-zero gradient."""
    import nerf
    c = C.build_case("train_rand_64_64")
    n_rays = 2048
    ro, rd, bg, tgt, idx = C.ray_subset(512, 512, c["frame"], n_rays, seed=31)
    t_rand, noise_c, u, noise_f = C.randoms(n_rays, 64, 64, seed=77)
    mc, mf = U.make_model(nerf, c["p_coarse"], gpu), U.make_model(nerf, c["p_fine"], gpu)
    opt = U.make_options(nerf, 64, 64, True, 0.1, chunksize=2048)
    ex, ed = U.encoders(nerf)
    nerf.set_mlp_precision(precision)

    def step(scale):
        for m in (mc, mf):
            m.zero_grad(set_to_none=True)
        latent = c["latent"].clone().to(gpu).requires_grad_(True)
        with torch.enable_grad(), U.injected_random([t_rand, u], [noise_c, noise_f]):
            out = nerf.run_one_iter_of_nerf(512, 512, None, mc, mf, ro.to(gpu), rd.to(gpu), opt, mode="train", encode_position_fn=ex,
                                            encode_direction_fn=ed, expressions=c["expr"].to(gpu), background_prior=bg.to(gpu),
                                            latent_code=latent)
            loss = O.train_loss(out[0], out[3], tgt.to(gpu), latent) * scale
            loss.backward()
        g = {f"{tag}.{k}": (None if v.grad is None else v.grad.clone()) for tag, m in (("coarse", mc), ("fine", mf))
             for k, v in m.named_parameters()}
        g["latent"] = latent.grad.clone()
        return float(loss.detach()), g
    l1, g1 = step(1.0)
    l1b, g1b = step(1.0)
    l2, g2 = step(2.0)
    assert l1 == l1b and l2 == 2 * l1
    for k, a in g1.items():
        if a is None:
            assert "layers_dir.3" in k and g2[k] is None                     # Quirk Q3
            continue
        assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0, k
        assert torch.equal(a, g1b[k]), k                                     # deterministic
        assert torch.equal(2 * a, g2[k]), k                                  # linear in the upstream gradient
