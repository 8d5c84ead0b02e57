"""Split-bf16 (3 x bf16 MFMA) forward: parity against the fp64 oracle / the exact-f32 kernel.  GPU only."""
import pytest
import torch

from oracle import cases as C
from oracle import nerface_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_rays,s", [(8, 64), (3, 7), (5, 192), (1, 1), (40, 192)])
def test_bf16x3_mlp_vs_oracle(hip_lib, gpu, n_rays, s):
    import nerf
    from nerf import ops
    c = C.build_case("eval_det_64_128")
    g = torch.Generator().manual_seed(5)
    ro, rd, _, _, _ = C.ray_subset(512, 512, 3, n_rays, 5)
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0]
    p = c["p_fine"]
    m = U.make_model(nerf, p, gpu)
    hw = m.hip_weights()
    cond = ops.paper_condition(hw.get(), c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    raw_b = ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, ro.to(gpu), rd.to(gpu), z.to(gpu)).cpu()
    raw_f = ops.paper_mlp_fwd(hw.get(), cond, ro.to(gpu), rd.to(gpu), z.to(gpu)).cpu()
    p64 = {k: v.double() for k, v in p.items()}
    ref = O.paper_mlp(p64, O.encode_points(ro.double(), rd.double(), z.double(), O.NEAR, O.FAR), c["expr"].double(),
                      c["latent"].double()).reshape(n_rays, s, 4)
    scale = ref.abs().amax(dim=(0, 1))
    eb = (raw_b.double() - ref).abs().amax(dim=(0, 1))
    ef = (raw_f.double() - ref).abs().amax(dim=(0, 1))
    print(f"bf16x3 max err {eb.tolist()}  f32 max err {ef.tolist()}  scale {scale.tolist()}")
    assert torch.all(eb <= 3e-4 * scale + 1e-5)


def test_bf16x3_psnr_gate(hip_lib, gpu):
    """The north-star gate with the split-bf16 inference kernel: |PSNR(ours,tgt) - PSNR(reference,tgt)| <= 1e-4 dB."""
    import nerf
    c = C.build_case("eval_det_64_128")
    ro, rd, bg, tgt, idx = C.ray_subset(512, 512, c["frame"], 1024, seed=99)
    c.update(n_rays=1024, ro=ro, rd=rd, bg=bg, tgt=tgt, idx=idx)
    ref = C.run_oracle(c)
    nerf.set_mlp_precision("bf16x3")
    try:
        out, *_ = U.run_product(nerf, c, gpu)
    finally:
        nerf.set_mlp_precision("f32")
    for k, name in ((0, "rgb_coarse"), (3, "rgb_fine")):
        p_ref, p_our = O.psnr(ref[k], tgt), O.psnr(out[k].cpu(), tgt)
        print(f"bf16x3 {name}: PSNR ref {p_ref:.6f} ours {p_our:.6f} |d|={abs(p_ref - p_our):.2e} dB, self-PSNR {O.psnr(out[k].cpu(), ref[k]):.1f} dB,"
              f" max|d rgb|={float((out[k].cpu() - ref[k]).abs().max()):.2e}")
        assert abs(p_ref - p_our) <= 1e-4


# coarse outputs: split-bf16 rounding (~2^-16 per layer) through one 11-layer MLP; fine outputs: plus the resampled-depth
# sensitivity shared with the exact-f32 path (tests/test_gpu_e2e.py: TOL)
TOL_B = dict(rgb_c=3e-5, rgb_f=5e-4, acc_c=1e-5, acc_f=1e-5, w_last=5e-4, disp_c=1e-4, disp_f=2e-3)


@pytest.mark.parametrize("name", list(C.CASES))
def test_bf16x3_against_golden_reference(hip_lib, gpu, name):
    import os
    import numpy as np
    import nerf
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", f"{name}.npz"))
    c = C.build_case(name)
    nerf.set_mlp_precision("bf16x3")
    try:
        out, *_ = U.run_product(nerf, c, gpu)
    finally:
        nerf.set_mlp_precision("f32")
    for n, t in zip(["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_last"], out):
        if n not in gold.files:
            assert t is None
            continue
        d = np.abs(t.cpu().numpy() - gold[n])
        print(f"[bf16x3 {name}] {n}: max|d|={d.max():.3e}")
        assert d.max() <= TOL_B[n], (n, d.max())


@pytest.mark.parametrize("frame,seed", [(11, 3), (57, 4), (90, 5)])
def test_bf16x3_psnr_gate_more_frames(hip_lib, gpu, frame, seed):
    """The 1e-4 dB gate on other frames / weight seeds / stochastic sampling (perturb on, injected draws)."""
    import nerf
    c = C.build_case("train_rand_64_64")
    n = 768
    ro, rd, bg, tgt, idx = C.ray_subset(512, 512, frame, n, seed=seed)
    expr, latent = O.frame_conditioning(frame)
    t_rand, noise_c, u, noise_f = C.randoms(n, 64, 128, seed=seed)
    c.update(frame=frame, n_rays=n, n_coarse=64, n_fine=128, ro=ro, rd=rd, bg=bg, tgt=tgt, idx=idx, expr=expr, latent=latent,
             stochastic=True, noise_std=0.0, t_rand=t_rand, u=u, noise_c=None, noise_f=None,
             p_coarse=O.init_paper_params(10 + seed), p_fine=O.init_paper_params(20 + seed))
    ref = C.run_oracle(c)
    nerf.set_mlp_precision("bf16x3")
    try:
        out, *_ = U.run_product(nerf, c, gpu)
    finally:
        nerf.set_mlp_precision("f32")
    for k in (0, 3):
        p_ref, p_our = O.psnr(ref[k], tgt), O.psnr(out[k].cpu(), tgt)
        print(f"frame {frame} output {k}: |dPSNR| = {abs(p_ref - p_our):.2e} dB, self-PSNR {O.psnr(out[k].cpu(), ref[k]):.1f} dB")
        assert abs(p_ref - p_our) <= 1e-4


def test_bf16x3_kernel_is_deterministic_under_load(hip_lib, gpu):
    """The LDS ring (DMA of stage g+3 in flight across the barrier, counted vmcnt) has no data race: 12 launches over a
    grid that oversubscribes the chip several times must be bit-identical, also against a tiny launch of the same rays."""
    import nerf
    from nerf import ops
    c = C.build_case("eval_det_64_128")
    g = torch.Generator().manual_seed(31)
    n_rays, s = 4096, 192
    ro, rd, _, _, _ = C.ray_subset(512, 512, 3, n_rays, 31)
    z = torch.sort(torch.rand((n_rays, s), generator=g) * 0.6 + 0.2, dim=-1)[0]
    m = U.make_model(nerf, c["p_fine"], gpu)
    hw = m.hip_weights()
    cond = ops.paper_condition(hw.get(), c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR)
    ro_d, rd_d, z_d = ro.to(gpu), rd.to(gpu), z.to(gpu)
    first = ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, ro_d, rd_d, z_d)
    for _ in range(11):
        again = ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, ro_d, rd_d, z_d)
        assert torch.equal(first, again)
    small = ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, ro_d[:5].contiguous(), rd_d[:5].contiguous(), z_d[:5].contiguous())
    assert torch.equal(small, first[:5])
    assert bool(torch.isfinite(first).all())
