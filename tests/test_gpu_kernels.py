"""Per-kernel parity of the HIP path (through the C ABI) against the CPU oracle.  GPU only."""
import numpy as np
import pytest
import torch

from oracle import cases as C
from oracle import nerface_oracle as O

pytestmark = pytest.mark.gpu

GOLD = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden")


def test_ray_bundle_bit_exact(hip_lib, gpu):
    import nerf
    pose = O.frame_pose(42)
    for (h, w) in [(37, 53), (512, 512), (1, 1)]:
        ro_o, rd_o = O.ray_bundle(h, w, O.INTRINSICS, pose)
        ro, rd = nerf.get_ray_bundle(h, w, O.INTRINSICS, pose.to(gpu))
        assert torch.equal(rd.cpu(), rd_o) and torch.equal(ro.cpu(), ro_o)
    g = np.load(f"{GOLD}/ray_bundle.npz")
    ro, rd = nerf.get_ray_bundle(37, 53, O.INTRINSICS, pose.to(gpu))
    assert np.array_equal(rd.cpu().numpy(), g["rd"]) and np.array_equal(ro.cpu().numpy(), g["ro"])
    _, rd_s = nerf.get_ray_bundle(24, 24, torch.tensor(138.88 * 24 / 100.0), pose[:3].to(gpu))   # scalar focal, 3x4 pose
    assert np.array_equal(rd_s.cpu().numpy(), g["rd_scalar"])


def test_ray_batch_equals_gathers_from_the_full_bundle(hip_lib, gpu):
    """The training-batch kernel (launcher's replacement for TR:302 + TR:325-330): rays of selected pixels must be the oracle's
    full bundle at these pixels bit for bit; target / background gathers must equal torch indexing; RGBA targets, duplicate
    and corner pixels, an empty selection and an out-of-range pixel (reported, not silently wrapped)."""
    import nerf
    pose = O.frame_pose(42)
    g = torch.Generator().manual_seed(5)
    for (h, w, ch, n) in [(512, 512, 3, 2048), (37, 53, 4, 300), (1, 1, 3, 2)]:
        ro_o, rd_o = O.ray_bundle(h, w, O.INTRINSICS, pose)
        sel = torch.stack((torch.randint(0, h, (n,), generator=g), torch.randint(0, w, (n,), generator=g)), dim=-1)
        sel[0] = torch.tensor([h - 1, w - 1])
        sel[-1] = sel[0]
        img, bg = torch.rand((h, w, ch), generator=g), torch.rand((h, w, 3), generator=g)
        ro, rd, tgt, b = nerf.get_ray_batch(h, w, O.INTRINSICS, pose.to(gpu), sel.to(gpu), img.to(gpu), bg.to(gpu), check=True)
        assert torch.equal(rd.cpu(), rd_o[sel[:, 0], sel[:, 1]]) and torch.equal(ro.cpu(), ro_o[sel[:, 0], sel[:, 1]])
        assert torch.equal(tgt.cpu(), img[sel[:, 0], sel[:, 1]]) and torch.equal(b.cpu(), bg[sel[:, 0], sel[:, 1]])
        ro2, rd2, t2, b2 = nerf.get_ray_batch(h, w, O.INTRINSICS, pose.to(gpu), sel.to(gpu))
        assert t2 is None and b2 is None and torch.equal(rd2, rd)
        flat = (sel[:, 0] * w + sel[:, 1]).to(gpu)                                   # flat pixel indices (what choose_rays returns)
        ro3, rd3, t3, b3 = nerf.get_ray_batch(h, w, O.INTRINSICS, pose.to(gpu), flat, img.to(gpu), bg.to(gpu), check=True)
        assert torch.equal(rd3, rd) and torch.equal(ro3, ro) and torch.equal(t3, tgt) and torch.equal(b3, b)
    ro, rd, tgt, b = nerf.get_ray_batch(8, 8, O.INTRINSICS, pose.to(gpu), torch.zeros((0, 2), dtype=torch.int64, device=gpu))
    assert ro.shape == (0, 3) and rd.shape == (0, 3)
    with pytest.raises(IndexError):
        nerf.get_ray_batch(8, 8, O.INTRINSICS, pose.to(gpu), torch.tensor([[3, 8]], device=gpu), check=True)
    with pytest.raises(IndexError):
        nerf.get_ray_batch(8, 8, O.INTRINSICS, pose.to(gpu), torch.tensor([64], device=gpu), check=True)
    with pytest.raises(IndexError):
        nerf.get_ray_batch(8, 8, O.INTRINSICS, pose.to(gpu), torch.tensor([-1], device=gpu), check=True)    # what K0 leaves when short
    with pytest.raises(ValueError):
        nerf.get_ray_batch(8, 8, O.INTRINSICS, pose.to(gpu), torch.tensor([[3.0, 1.0]], device=gpu))


@pytest.mark.parametrize("nc", [64, 5, 1, 192])
def test_sample_coarse_bit_exact(hip_lib, gpu, nc):
    from nerf import ops
    n = 33
    t_rand = torch.rand((n, nc), generator=torch.Generator().manual_seed(3))
    z = ops.sample_coarse(n, nc, O.NEAR, O.FAR, gpu, None)
    if nc > 1:
        assert torch.equal(z.cpu(), O.coarse_z(n, O.NEAR, O.FAR, nc, None))
        zr = ops.sample_coarse(n, nc, O.NEAR, O.FAR, gpu, t_rand.to(gpu))
        assert torch.equal(zr.cpu(), O.coarse_z(n, O.NEAR, O.FAR, nc, t_rand))
        # lindisp (T:65-66): linear in disparity, same bit-exact contract; also with the tiny path's depth range
        for near, far in ((O.NEAR, O.FAR), (2.0, 6.0)):
            zl = ops.sample_coarse(n, nc, near, far, gpu, None, lindisp=True)
            assert torch.equal(zl.cpu(), O.coarse_z(n, near, far, nc, None, lindisp=True))
            zlr = ops.sample_coarse(n, nc, near, far, gpu, t_rand.to(gpu), lindisp=True)
            assert torch.equal(zlr.cpu(), O.coarse_z(n, near, far, nc, t_rand, lindisp=True))


def test_posenc(hip_lib, gpu):
    import nerf
    g = np.load(f"{GOLD}/pe_pdf.npz")
    x = torch.from_numpy(g["x"])
    pe10 = nerf.positional_encoding(x.to(gpu), 10, True).cpu()
    pe4 = nerf.get_embedding_function(4, False)(x.to(gpu)).cpu()
    assert pe10.shape == (33, 63) and pe4.shape == (33, 24)
    # sin/cos of arguments up to ~400 rad: device libm vs host libm, a few ulp
    assert np.abs(pe10.numpy() - g["pe10"]).max() < 2e-6
    assert np.abs(pe4.numpy() - g["pe4"]).max() < 2e-6
    assert torch.equal(pe10[:, :3], x)
    assert nerf.positional_encoding(torch.zeros((0, 3), device=gpu), 10, True).shape == (0, 63)


def _mlp_inputs(n_rays, s, seed):
    c = C.build_case("eval_det_64_128")
    g = torch.Generator().manual_seed(seed)
    ro, rd, _, _, _ = C.ray_subset(512, 512, 3, n_rays, seed)
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0]
    return c, ro, rd, z


@pytest.mark.parametrize("n_rays,s", [(8, 64), (3, 7), (5, 192), (1, 1)])
def test_paper_mlp_fwd(hip_lib, gpu, n_rays, s):
    import nerf
    from nerf import ops
    c, ro, rd, z = _mlp_inputs(n_rays, s, 5)
    p = c["p_fine"]
    m = nerf.models.ConditionalBlendshapePaperNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True,
                                                        include_input_dir=False)
    m.load_state_dict(p)
    m.to(gpu)
    pk = m.hip_weights().get()
    cond = ops.paper_condition(pk, c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    raw = ops.paper_mlp_fwd(pk, cond, ro.to(gpu), rd.to(gpu), z.to(gpu)).cpu()
    p64 = {k: v.double() for k, v in p.items()}
    ref = O.paper_mlp(p64, O.encode_points(ro.double(), rd.double(), z.double(), O.NEAR, O.FAR), c["expr"].double(),
                      c["latent"].double()).reshape(n_rays, s, 4)
    ref32 = O.paper_mlp(p, O.encode_points(ro, rd, z, O.NEAR, O.FAR), c["expr"], c["latent"]).reshape(n_rays, s, 4)
    err = (raw.double() - ref).abs()
    err32 = (ref32.double() - ref).abs()
    scale = ref.abs().amax(dim=(0, 1))
    print("mlp max err vs fp64:", err.amax(dim=(0, 1)).tolist(), "fp32-oracle err:", err32.amax(dim=(0, 1)).tolist(),
          "scale", scale.tolist())
    # fp32 accumulation-order noise: bounded relative to each channel's magnitude (sigma is boosted x1000)
    assert torch.all(err.amax(dim=(0, 1)) <= 2e-5 * scale + 2e-5)


def test_paper_mlp_fwd_persistent_grid_is_launch_invariant(hip_lib, gpu):
    """The exact-f32 inference kernel runs a persistent grid (one workgroup per CU walks the 128-point blocks with the grid's stride).
    Size-independent property: a launch with more blocks than the grid (700 x 67 = 46900 points = 367 blocks, the last one partial)
    must give, bit for bit, what two smaller launches over the same rays give -- a point's result cannot depend on the block that
    computed it, nor on which pass of the workgroup's loop that was."""
    import nerf
    from nerf import ops
    c, ro, rd, z = _mlp_inputs(700, 67, 23)
    m = nerf.models.ConditionalBlendshapePaperNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True,
                                                        include_input_dir=False)
    m.load_state_dict(c["p_fine"])
    m.to(gpu)
    pk = m.hip_weights().get()
    cond = ops.paper_condition(pk, c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    ro, rd, z = ro.to(gpu), rd.to(gpu), z.to(gpu)
    big = ops.paper_mlp_fwd(pk, cond, ro, rd, z)
    parts = [ops.paper_mlp_fwd(pk, cond, ro[a:b].contiguous(), rd[a:b].contiguous(), z[a:b].contiguous()) for a, b in ((0, 301), (301, 700))]
    assert torch.isfinite(big).all()
    assert torch.equal(big, torch.cat(parts, dim=0))
    # and the training forward (not persistent, its own K loops) returns the same raw values
    raw_t, _ = ops.paper_mlp_fwd_train(pk, cond, ro, rd, z)
    assert torch.equal(raw_t, big)


@pytest.mark.parametrize("n_rays,s,bgflag,noisy", [(9, 64, True, False), (4, 192, True, True), (7, 5, False, False), (3, 130, True, True)])
def test_volume_render_fwd(hip_lib, gpu, n_rays, s, bgflag, noisy):
    from nerf import ops
    g = torch.Generator().manual_seed(11)
    raw = torch.randn((n_rays, s, 4), generator=g) * 3.0
    raw[..., 3] = raw[..., 3] * 10
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0]
    rd = torch.randn((n_rays, 3), generator=g)
    bg = torch.rand((n_rays, 3), generator=g) if bgflag else None
    noise = torch.randn((n_rays, s), generator=g) * 0.1 if noisy else None
    raw_o = raw.clone()
    if bg is not None:
        raw_o[:, -1, :3] = bg
    rgb_o, disp_o, acc_o, w_o = O.volume_render(raw_o.double(), z.double(), rd.double(), None if noise is None else noise.double(),
                                                has_background=bgflag)
    dv = lambda t: None if t is None else t.to(gpu)
    rgb, disp, acc, w = ops.volume_render_fwd(dv(raw), dv(z), dv(rd), dv(noise), dv(bg))
    assert (w.cpu().double() - w_o).abs().max() < 2e-6
    assert (rgb.cpu().double() - rgb_o).abs().max() < 5e-6
    assert (acc.cpu().double() - acc_o).abs().max() < 5e-6
    assert ((disp.cpu().double() - disp_o).abs() / disp_o.abs()).max() < 1e-5
    if bgflag:
        assert (acc.cpu() - 1).abs().max() < 1e-5           # Q5: alpha_last == 1 -> acc == 1


def _ulp_diff(a, b):
    """Distance in units in the last place between two float32 arrays (same sign region)."""
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


def test_sample_pdf_bit_exact(hip_lib, gpu):
    """K6 against the reference's own sample_pdf_2 outputs (tests/golden/pe_pdf.npz, H:344-387): the CDF table and the
    searchsorted indices are IDENTICAL to torch-CPU's (nf_build_cdf reproduces ATen's row-sum order and the sequential
    double cumsum), hence the samples are too -- no tolerance clause.  Covers the edge rows (all-zero weights, one spike,
    uniform, u = 0 / 1 / 0.999999, det mode) and table widths 4 .. 191 (scalar and vector row-sum paths)."""
    import nerf
    from nerf import ops
    g = np.load(f"{GOLD}/pe_pdf.npz")
    bins, w, u = (torch.from_numpy(g[k]) for k in ("bins", "w", "u"))
    zs, inds, cdf = ops.sample_pdf(bins.to(gpu), w.to(gpu), 128, u.to(gpu), want_table=True)
    assert np.array_equal(cdf.cpu().numpy(), g["cdf"])
    assert np.array_equal(inds.cpu().numpy(), g["inds_rand"])
    assert _ulp_diff(zs.cpu().numpy(), g["zs_rand"]).max() <= 1
    zd, inds_d, _ = ops.sample_pdf(bins.to(gpu), w.to(gpu), 128, None, want_table=True)
    assert np.array_equal(inds_d.cpu().numpy(), g["inds_det"])
    assert _ulp_diff(zd.cpu().numpy(), g["zs_det"]).max() <= 1
    assert np.array_equal(nerf.sample_pdf_2(bins.to(gpu), w.to(gpu), 128, det=True).cpu().numpy(), zd.cpu().numpy())
    for nb in (4, 10, 18, 63, 191):
        b, ww, uu = (torch.from_numpy(g[f"b{nb}_{k}"]).to(gpu) for k in ("bins", "w", "u"))
        z2, i2, c2 = ops.sample_pdf(b, ww, 64, uu, want_table=True)
        assert np.array_equal(c2.cpu().numpy(), g[f"b{nb}_cdf"]), nb
        assert np.array_equal(i2.cpu().numpy(), g[f"b{nb}_inds"]), nb
        assert _ulp_diff(z2.cpu().numpy(), g[f"b{nb}_zs"]).max() <= 1, nb
    # live oracle on this host (its torch build may sum rows in another vector width): indices may differ only where u sits
    # within rounding of a knot, so compare the tables to 2 ulp and the indices away from the knots exactly
    gg = torch.Generator().manual_seed(9)
    b = torch.sort(torch.rand((300, 63), generator=gg) * 0.6 + 0.2, dim=-1)[0]
    ww = torch.rand((300, 62), generator=gg) ** 4
    uu = torch.rand((300, 128), generator=gg)
    t = {}
    z_o = O.sample_pdf(b, ww, 128, uu, table=t)
    z3, i3, c3 = ops.sample_pdf(b.to(gpu), ww.to(gpu), 128, uu.to(gpu), want_table=True)
    assert (c3.cpu() - t["cdf"]).abs().max() <= 2.4e-7
    near_knot = ((uu[:, :, None] - t["cdf"][:, None, :]).abs() <= 4e-7).any(-1)
    assert torch.equal(i3.cpu().long()[~near_knot], t["inds"][~near_knot])
    assert float(near_knot.float().mean()) < 1e-3
    assert (z3.cpu() - z_o)[~near_knot].abs().max() < 2e-6


def test_sort_rows(hip_lib, gpu):
    from nerf import ops
    x = torch.rand((11, 192), generator=torch.Generator().manual_seed(1))
    assert torch.equal(ops.sort_rows(x.to(gpu)).cpu(), torch.sort(x, dim=-1)[0])
    x = torch.rand((5, 13), generator=torch.Generator().manual_seed(2))
    assert torch.equal(ops.sort_rows(x.to(gpu)).cpu(), torch.sort(x, dim=-1)[0])


def test_resample_merge_fast_path_is_bit_identical(hip_lib, gpu):
    """k_resample_merge_small (shipped sizes: wave scan in double behind an exactness guard, register sort of the samples, merge by
    rank) against the separately verified kernels -- z_samples must EQUAL k_sample_pdf's (whose table / indices are array_equal to
    torch-CPU's, test_sample_pdf_bit_exact) and z_fine must EQUAL k_sort_rows(cat) -- bit for bit, at 64+128, 64+64, ragged sizes, the
    golden edge rows, rows whose pdf fails the exactness guard (dynamic range > 2^28: the one-lane double loop runs) and rows whose
    coarse depths are not ascending (the in-kernel full sort runs)."""
    from nerf import ops
    g = torch.Generator().manual_seed(21)

    def check(z_c, w_c, nf, u):
        z_f, z_s = ops.resample_merge(z_c.to(gpu), w_c.to(gpu), nf, None if u is None else u.to(gpu), want_samples=True)
        bins = 0.5 * (z_c[:, 1:] + z_c[:, :-1])
        want_s = ops.sample_pdf(bins.to(gpu), w_c[:, 1:-1].contiguous().to(gpu), nf, None if u is None else u.to(gpu))
        assert torch.equal(z_s, want_s)
        want_f = ops.sort_rows(torch.cat((z_c.to(gpu), want_s), dim=-1).contiguous())
        assert torch.equal(z_f, want_f)
        assert torch.equal(z_f.cpu(), torch.sort(torch.cat((z_c, want_s.cpu()), -1), -1)[0])

    for n_rays, nc, nf in ((4099, 64, 128), (2048, 64, 64), (37, 5, 7), (130, 128, 128), (9, 3, 1), (64, 33, 100)):
        z_c = torch.sort(torch.rand((n_rays, nc), generator=g) * 0.6 + 0.2, dim=-1)[0]
        w_c = torch.rand((n_rays, nc), generator=g) ** 6                      # a few spikes, many near-zero weights
        w_c[::7] = 0.0                                                         # all-zero rows (uniform pdf)
        w_c[1::7, nc // 2] = 1.0
        u = torch.rand((n_rays, nf), generator=g)
        if nf >= 3:
            u[0, :3] = torch.tensor([0.0, 1.0, 0.999999])
        check(z_c, w_c, nf, u)
        check(z_c, w_c, nf, None)                                              # det: linspace abscissae
    # exactness guard fails: one weight of 1e7 makes the other pdf entries ~1e-12 < 2^-28 -> sequential double cumsum
    z_c = torch.sort(torch.rand((50, 64), generator=g) * 0.6 + 0.2, dim=-1)[0]
    w_c = torch.rand((50, 64), generator=g)
    w_c[:, 20] = 1e7
    w_c[5] = -2e-5                                                             # negative weights: pdf entries of both signs
    check(z_c, w_c, 128, torch.rand((50, 128), generator=g))
    # coarse depths not ascending (no caller on the hot path does this): the result is still sort(cat)
    z_c = torch.rand((20, 64), generator=g) * 0.6 + 0.2
    check(z_c, torch.rand((20, 64), generator=g), 128, torch.rand((20, 128), generator=g))
    # golden edge rows of the reference's own sample_pdf_2 fixture, through the fused kernel (bins -> depths: z = bins padded by one)
    gd = np.load(f"{GOLD}/pe_pdf.npz")
    bins, w, u = (torch.from_numpy(gd[k]) for k in ("bins", "w", "u"))
    z_c = torch.cat((bins[:, :1], bins), dim=-1)                               # 64 depths whose interior midpoints are not the bins, fine
    check(z_c.contiguous(), torch.cat((w[:, :1], w, w[:, :1]), dim=-1).contiguous(), 128, u)
    # NaN samples (NaN u / NaN weights: a diverged model) stay NaN in z_fine: the register sort is a v_min network, which would
    # swallow them and duplicate a neighbour -- such rows take the compare-swap path, whose comparison is a total order with NaNs last
    # (ADVICE r04): z_fine equals torch.sort(cat(depths, samples)), NaNs included.
    for nc, nf in ((64, 128), (64, 64), (33, 100)):
        z_c = torch.sort(torch.rand((70, nc), generator=g) * 0.6 + 0.2, dim=-1)[0]
        w_c = torch.rand((70, nc), generator=g)
        u = torch.rand((70, nf), generator=g)
        u[5, 7] = float("nan")
        u[6, ::3] = float("nan")
        w_c[9, nc // 3] = float("nan")
        z_f, z_s = ops.resample_merge(z_c.to(gpu), w_c.to(gpu), nf, u.to(gpu), want_samples=True)
        want_s = ops.sample_pdf((0.5 * (z_c[:, 1:] + z_c[:, :-1])).to(gpu), w_c[:, 1:-1].contiguous().to(gpu), nf, u.to(gpu))
        assert torch.equal(torch.isnan(z_s), torch.isnan(want_s)) and torch.equal(torch.nan_to_num(z_s, nan=-1.0), torch.nan_to_num(want_s, nan=-1.0))
        assert int(torch.isnan(z_s[5]).sum()) == 1 and int(torch.isnan(z_s[6]).sum()) == len(range(0, nf, 3)) and bool(torch.isnan(z_s[9]).any())
        cat = torch.cat((z_c.to(gpu), z_s), dim=-1)
        want_f = torch.sort(cat.cpu(), dim=-1)[0]                             # torch's order: ascending, NaNs last
        for got in (z_f, ops.sort_rows(cat.contiguous())):                    # the fused kernel and the stand-alone row sort
            assert torch.equal(torch.isnan(got).cpu(), torch.isnan(want_f))
            assert torch.equal(torch.nan_to_num(got, nan=-1.0).cpu(), torch.nan_to_num(want_f, nan=-1.0))
    # ... and through the general kernel (sizes beyond the register path) with +inf among the values: inf sorts before the NaNs
    z_c = torch.sort(torch.rand((6, 140), generator=g) * 0.6 + 0.2, dim=-1)[0]
    w_c = torch.rand((6, 140), generator=g)
    u = torch.rand((6, 100), generator=g)
    u[2, 5] = float("nan")
    z_c[3, -1] = float("inf")
    z_f, z_s = ops.resample_merge(z_c.to(gpu), w_c.to(gpu), 100, u.to(gpu), want_samples=True)
    want_f = torch.sort(torch.cat((z_c, z_s.cpu()), -1), -1)[0]
    assert torch.equal(torch.isnan(z_f).cpu(), torch.isnan(want_f)) and torch.equal(torch.nan_to_num(z_f, nan=-1.0).cpu(), torch.nan_to_num(want_f, nan=-1.0))
    assert int(torch.isnan(z_f[2]).sum()) == 1 and bool(torch.isinf(z_f[3]).any())


def test_resample_merge(hip_lib, gpu):
    """K6+K7 fused on the oracle's own coarse weights: identical inputs -> the resampled depths and the merged, sorted
    fine depths equal the oracle's (the table is reproduced exactly; <= 1 ulp allowed for the interpolation)."""
    from nerf import ops
    c = C.build_case("train_rand_64_64")
    st = {}
    C.run_oracle(c, st)
    z_f, z_s = ops.resample_merge(st["z_c"].to(gpu), st["w_c"].to(gpu), 64, c["u"].to(gpu), want_samples=True)
    t = {}
    z_mid = 0.5 * (st["z_c"][:, 1:] + st["z_c"][:, :-1])
    O.sample_pdf(z_mid, st["w_c"][:, 1:-1], 64, c["u"], table=t)
    near_knot = ((c["u"][:, :, None] - t["cdf"][:, None, :]).abs() <= 4e-7).any(-1)
    d = (z_s.cpu() - st["z_samples"]).abs()
    assert d[~near_knot].max() < 2e-6 and float(near_knot.float().mean()) < 1e-3
    assert torch.all(z_f[:, 1:] >= z_f[:, :-1])
    if not bool(near_knot.any()):
        assert (z_f.cpu() - st["z_f"]).abs().max() < 2e-6


def test_model_forward_and_run_network_on_encoded_inputs(hip_lib, gpu):
    """model(x87, expr, latent) and nerf.run_network (the reference's unfused call chain, T:9-33) against the oracle."""
    import nerf
    from tests import util as U
    c, ro, rd, z = _mlp_inputs(6, 37, 8)
    p = c["p_coarse"]
    m = U.make_model(nerf, p, gpu)
    x87 = O.encode_points(ro, rd, z, O.NEAR, O.FAR)
    with torch.no_grad():
        out = m(x87.to(gpu), c["expr"].to(gpu), c["latent"].to(gpu)).cpu()
    p64 = {k: v.double() for k, v in p.items()}
    ref = O.paper_mlp(p64, x87.double(), c["expr"].double(), c["latent"].double())
    scale = ref.abs().amax(dim=0)
    assert torch.all((out.double() - ref).abs().amax(dim=0) <= 2e-5 * scale + 2e-5)
    # run_network: same call as the reference's predict_and_render_radiance makes (T:84-93)
    ex, ed = U.encoders(nerf)
    pts = (ro[:, None, :] + rd[:, None, :] * z[:, :, None]).to(gpu)
    ray_batch = torch.cat((ro, rd, torch.full((6, 1), O.NEAR), torch.full((6, 1), O.FAR)), dim=-1).to(gpu)
    with torch.no_grad():
        rf = nerf.run_network(m, pts, ray_batch, 100, ex, ed, c["expr"].to(gpu), c["latent"].to(gpu)).cpu()
    assert rf.shape == (6, 37, 4)
    assert torch.all((rf.reshape(-1, 4).double() - ref).abs().amax(dim=0) <= 3e-5 * scale + 3e-5)
    with pytest.raises(NotImplementedError):
        m(x87.to(gpu), c["expr"].to(gpu), c["latent"].to(gpu).requires_grad_(True))


def test_empty_inputs(hip_lib, gpu):
    """Zero rays through every stage: shapes come back empty, nothing is launched, nothing raises."""
    import nerf
    from nerf import ops
    c = C.build_case("eval_det_64_128")
    m = nerf.models.ConditionalBlendshapePaperNeRFModel(num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True,
                                                        include_input_dir=False)
    m.load_state_dict(c["p_fine"])
    m.to(gpu)
    e = lambda *shape: torch.empty(shape, device=gpu)
    z = ops.sample_coarse(0, 64, O.NEAR, O.FAR, gpu, None)
    assert z.shape == (0, 64)
    hw = m.hip_weights()
    cond = ops.paper_condition(hw.get(), c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    for raw in (ops.paper_mlp_fwd(hw.get(), cond, e(0, 3), e(0, 3), z), ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, e(0, 3), e(0, 3), z),
                ops.paper_mlp_fwd_f16(hw.get_f16(), cond, e(0, 3), e(0, 3), z)):
        assert raw.shape == (0, 64, 4)
    rgb, disp, acc, w = ops.volume_render_fwd(raw, z, e(0, 3), None, e(0, 3))
    assert rgb.shape == (0, 3) and w.shape == (0, 64)
    assert ops.resample_merge(z, w, 128, None).shape == (0, 192)
    assert ops.sample_pdf(e(0, 63), e(0, 62), 128, None).shape == (0, 128)
    assert ops.sort_rows(e(0, 17)).shape == (0, 17)
    opt = __import__("tests.util", fromlist=["x"]).make_options(nerf, 64, 128, False, 0.0)
    ex, ed = __import__("tests.util", fromlist=["x"]).encoders(nerf)
    with torch.no_grad():
        out = nerf.run_one_iter_of_nerf(512, 512, None, m, m, e(0, 3), e(0, 3), opt, mode="train", encode_position_fn=ex, encode_direction_fn=ed,
                                        expressions=c["expr"].to(gpu), background_prior=e(0, 3), latent_code=c["latent"].to(gpu))
    assert out == ()          # as the reference: no ray chunks -> zip(*[]) -> an empty tuple (T:229-290)


def test_weighted_choice_is_the_n_smallest_exponential_keys(hip_lib, gpu):
    """K0 (TR:320-322 on the device): the radix select must return exactly the n items with the smallest keys -log(1-u)/w --
    checked against a float64 evaluation of the keys (boundary items may swap within fp32 rounding of the key) --, distinct,
    in range, never an item of weight 0; ties in the key, n = 1, n = all positive items, too few positive items."""
    from nerf import ops
    g = torch.Generator().manual_seed(101)
    for n_items, n in ((262144, 2048), (5000, 4948), (5000, 4900), (1000, 1), (777, 300)):     # 4948 = every item of positive weight
        w = torch.full((n_items,), 0.1)
        w[n_items // 5: n_items // 2] = 0.9                                   # the trainer's two-level importance map
        w[::97] = 0.0
        w = w / w.sum()
        u = torch.rand(n_items, generator=g)
        idx = ops.weighted_choice(w.to(gpu), n, u.to(gpu), check=True).cpu()
        assert idx.shape == (n,) and idx.dtype == torch.int64
        assert bool((idx[1:] > idx[:-1]).all()) or n == 1                     # ascending (deterministic batch order)
        assert int(idx.min()) >= 0 and int(idx.max()) < n_items and len(set(idx.tolist())) == n
        assert bool((w[idx] > 0).all())
        key = -torch.log1p(-u.double()) / w.double()
        key[w == 0] = float("inf")
        chosen = torch.zeros(n_items, dtype=torch.bool)
        chosen[idx] = True
        assert float(key[chosen].max()) <= float(key[~chosen].min()) * (1 + 1e-5)
    # ties: identical weights and identical random numbers -> any n of them
    idx = ops.weighted_choice(torch.ones(4096, device=gpu), 100, torch.full((4096,), 0.25, device=gpu), check=True).cpu()
    assert len(set(idx.tolist())) == 100
    # fewer positive weights than requested: numpy raises ValueError; without the check the missing entries stay -1
    w = torch.zeros(1000)
    w[:10] = 1.0
    with pytest.raises(ValueError):
        ops.weighted_choice(w.to(gpu), 20, check=True)
    idx = ops.weighted_choice(w.to(gpu), 20).cpu()
    assert sorted(idx[idx >= 0].tolist()) == list(range(10)) and int((idx < 0).sum()) == 10
    assert ops.weighted_choice(w.to(gpu), 0).numel() == 0


def test_weighted_choice_follows_the_importance_map(hip_lib, gpu):
    """Distribution: with the trainer's map (p = 0.9 inside the bounding box, TR:230-239) the share of chosen pixels inside the
    box must match torch.multinomial's (same sampling scheme, independent implementation) within 4 standard errors, and every
    call must draw fresh numbers from torch's generator (reproducible under manual_seed)."""
    import nerf
    H, W, n = 512, 512, 2048
    m = torch.full((H, W), 0.1)
    m[150:400, 120:380] = 0.9
    p = (m / m.sum()).reshape(-1).to(gpu)
    inside = (m.reshape(-1) > 0.5).to(gpu)
    torch.manual_seed(5)
    a = torch.stack([inside[nerf.choose_rays(p, n)].float().mean() for _ in range(200)])
    b = torch.stack([inside[torch.multinomial(p, n, replacement=False)].float().mean() for _ in range(200)])
    se = float((a.var() / 200 + b.var() / 200).sqrt())
    print(f"share inside the box: K0 {float(a.mean()):.4f}, torch.multinomial {float(b.mean()):.4f}, standard error {se:.5f}")
    assert abs(float(a.mean() - b.mean())) < 4 * se
    torch.manual_seed(77)
    i1 = nerf.choose_rays(p, n)
    i2 = nerf.choose_rays(p, n)
    torch.manual_seed(77)
    j1 = nerf.choose_rays(p, n)
    assert torch.equal(i1, j1) and not torch.equal(i1, i2)


def test_weighted_choice_breaks_ties_by_index(hip_lib, gpu):
    """K0 with many equal keys (constant weights, u from a 128-value grid: every key is shared by ~512 pixels -- far more than a real
    draw's 1.5 % chance of ONE shared threshold key): the batch is the
    n smallest (key, index) pairs -- ties go to the LOWEST indices, so the draw is a function of the seed, not of the order in
    which workgroups happen to run -- and repeating the call reproduces it element for element."""
    from nerf import ops
    n_items, n = 256 * 256, 2000
    g = torch.Generator().manual_seed(3)
    u = (torch.randint(0, 128, (n_items,), generator=g).float() + 0.5) / 128.0
    w = torch.full((n_items,), 1.0 / n_items)
    want = torch.sort(torch.argsort(u.double() * n_items + torch.arange(n_items).double() / n_items, stable=True)[:n])[0]
    got = [ops.weighted_choice(w.to(gpu), n, u=u.to(gpu)).cpu() for _ in range(3)]
    assert torch.equal(got[0], want), int((got[0] != want).sum())
    assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[2])


def test_one_launch_adam_follows_torch_adam(hip_lib, gpu):
    """nerf.optim.Adam (nf_adam_step: every tensor of a step in one launch) against torch.optim.Adam on the same parameters and
    gradients: ragged sizes (the vector path and the scalar tail), a parameter without gradient (skipped, like layers_dir.3), a
    learning-rate change between steps (the trainer decays lr every iteration), and the state layout -- a state_dict saved by one
    loads into the other and both continue in step."""
    import nerf
    g = torch.Generator().manual_seed(5)
    shapes = [(256, 171), (256,), (3, 128), (1,), (1000, 32), (7, 5), (128, 280)]
    base = [torch.randn(s, generator=g) * 0.1 for s in shapes]
    mk = lambda: [torch.nn.Parameter(b.clone().to(gpu)) for b in base]
    pa, pb = mk(), mk()
    oa = nerf.optim.Adam(pa, lr=5e-4)
    ob = torch.optim.Adam(pb, lr=5e-4)
    for it in range(6):
        lr = 5e-4 * (0.1 ** (it / 250000.0)) if it else 5e-4
        for o in (oa, ob):
            for grp in o.param_groups:
                grp["lr"] = lr
        for k, (a, b) in enumerate(zip(pa, pb)):
            if k == 3 and it % 2:                                   # a tensor that sometimes has no gradient
                a.grad = b.grad = None
                continue
            gr = (torch.randn(shapes[k], generator=g) * (10.0 ** -(k % 4))).to(gpu)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
        if it == 2:                                                 # cross-load the optimizer state (checkpoint compatibility, both ways)
            sa, sb = oa.state_dict(), ob.state_dict()
            assert set(sa["state"][0]) == set(sb["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
            oa.load_state_dict(sb)
            ob.load_state_dict(sa)
    for k, (a, b) in enumerate(zip(pa, pb)):
        err = float((a - b).abs().max() / (b.abs().max() + 1e-12))
        assert err < 2e-6, (k, err)
        assert float((a.detach().cpu() - base[k]).abs().max()) > 0         # the parameters did move
    assert float(oa.state[pa[3]]["step"]) == 3.0 and float(oa.state[pa[0]]["step"]) == 6.0
    with pytest.raises(NotImplementedError):
        nerf.optim.Adam(mk(), lr=1e-3, weight_decay=0.1)
    # a checkpoint of the reference's torch versions holds `step` as a Python int; a fused / capturable one on the device:
    # both resume, in step with torch.optim.Adam; version counters move (caches keyed on them see the update); an empty tensor is skipped
    for conv in (lambda t: int(t.item()), lambda t: t.to(gpu)):
        import copy
        sd = copy.deepcopy(ob.state_dict())                          # (state_dict() hands out the optimizer's own per-parameter dicts)
        for st in sd["state"].values():
            st["step"] = conv(st["step"])
        pc = [torch.nn.Parameter(b.detach().clone()) for b in pb] + [torch.nn.Parameter(torch.zeros(0, device=gpu))]
        oc = nerf.optim.Adam(pc[:-1], lr=5e-4)
        oc.load_state_dict(sd)
        oc.add_param_group({"params": [pc[-1]]})
        before = [p._version for p in pc[:-1]]
        for k, (c_, b) in enumerate(zip(pc, pb)):
            gr = (torch.randn(shapes[k], generator=g) * 0.01).to(gpu)
            c_.grad, b.grad = gr.clone(), gr.clone()
        pc[-1].grad = torch.zeros(0, device=gpu)
        oc.step()
        ob.step()
        assert all(p._version > v for p, v in zip(pc[:-1], before))
        assert float(oc.state[pc[0]]["step"]) == float(ob.state[pb[0]]["step"]) and not oc.state[pc[0]]["step"].is_cuda
        for k, (c_, b) in enumerate(zip(pc, pb)):
            assert float((c_ - b).abs().max() / (b.abs().max() + 1e-12)) < 2e-6, k


def test_training_loss_equals_the_torch_expression(hip_lib, gpu):
    """nerf.training_loss (TR:355-387 in two launches) against the trainer's own torch expression on the device: value and the three
    gradients, with a fine map and without, a zero latent code (ATen's norm_backward gives 0 there), strided colour maps."""
    import nerf
    g = torch.Generator().manual_seed(5)
    for n, with_fine, zero_lat, strided in ((2048, True, False, False), (2048, True, True, True), (37, False, False, False), (1, True, False, True)):
        c4 = torch.rand((n, 4), generator=g).to(gpu)
        f4 = torch.rand((n, 4), generator=g).to(gpu)
        tgt = torch.rand((n, 3), generator=g).to(gpu)
        lat = (torch.zeros(32) if zero_lat else 0.1 * torch.randn(32, generator=g)).to(gpu)
        leaves = [t.clone().requires_grad_(True) for t in (c4, f4, lat)]
        a, b, l = leaves
        rc, rf = (a[..., :3], b[..., :3]) if strided else (a[..., :3].contiguous(), b[..., :3].contiguous())
        want = torch.nn.functional.mse_loss(rc, tgt) + (torch.nn.functional.mse_loss(rf, tgt) if with_fine else 0.0) + 10 * (torch.norm(l) * 0.0005)
        (3.0 * want).backward()
        want_g = [x.grad.clone() for x in leaves if x.grad is not None]
        leaves2 = [t.clone().requires_grad_(True) for t in (c4, f4, lat)]
        a2, b2, l2 = leaves2
        rc2, rf2 = (a2[..., :3], b2[..., :3]) if strided else (a2[..., :3].contiguous(), b2[..., :3].contiguous())
        got, parts = nerf.training_loss(rc2, rf2 if with_fine else None, tgt, l2)
        assert got.shape == () and not parts.requires_grad and parts.shape == (7,)
        (3.0 * got).backward()
        got_g = [x.grad.clone() for x in leaves2 if x.grad is not None]
        assert abs(float(got) - float(want)) <= 2e-7 * abs(float(want)) + 1e-12
        p = parts.cpu()
        assert abs(float(p[0]) - float(got)) == 0.0
        assert abs(float(p[1]) - float(torch.nn.functional.mse_loss(rc, tgt))) < 1e-7
        assert abs(float(p[3]) - float(torch.norm(lat) * 0.0005)) < 1e-9
        assert abs(float(p[5]) - float(-10.0 * torch.log10(p[4].clamp_min(1e-20)))) < 1e-4
        assert len(got_g) == len(want_g)
        for x, y in zip(got_g, want_g):
            assert x.shape == y.shape and torch.allclose(x, y, rtol=2e-6, atol=1e-12), float((x - y).abs().max())
        if zero_lat:
            assert float(leaves2[2].grad.abs().max()) == 0.0
