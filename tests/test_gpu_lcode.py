"""Second model family (ConditionalBlendshapeLearnableCodeNeRFModel): inference parity against the reference's golden
output and the fp64 oracle.  GPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import cases as C
from oracle import nerface_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = dict(rgb_c=3e-6, rgb_f=5e-4, acc_c=1e-5, acc_f=1e-5, w_last=5e-4, disp_c=1e-5, disp_f=2e-3)


def lmodel(nerf, params, device):
    m = nerf.models.ConditionalBlendshapeLearnableCodeNeRFModel(
        num_encoding_fn_xyz=10, num_encoding_fn_dir=4, include_input_xyz=True, include_input_dir=False, use_viewdirs=True,
        num_layers=4, hidden_size=256, include_expression=True)
    assert list(m.state_dict().keys()) == O.LCODE_KEYS
    m.load_state_dict(params)
    return m.to(device)


def test_lcode_eval_against_golden_reference(hip_lib, gpu):
    import nerf
    gold = np.load(os.path.join(GOLD, "lcode_eval_det_64_128.npz"))
    c = C.build_case("eval_det_64_128")
    mc, mf = lmodel(nerf, O.init_lcode_params(5), gpu), lmodel(nerf, O.init_lcode_params(6), gpu)
    assert mc.fused_supported()
    opt = U.make_options(nerf, 64, 128, False, 0.0)
    ex, ed = U.encoders(nerf)
    with torch.no_grad():
        out = nerf.run_one_iter_of_nerf(512, 512, None, mc, mf, c["ro"].to(gpu), c["rd"].to(gpu), opt, mode="train",
                                        encode_position_fn=ex, encode_direction_fn=ed, expressions=c["expr"].to(gpu),
                                        background_prior=c["bg"].to(gpu), latent_code=c["latent"].to(gpu))
    for n, t in zip(["rgb_c", "disp_c", "acc_c", "rgb_f", "disp_f", "acc_f", "w_last"], out):
        d = np.abs(t.cpu().numpy() - gold[n])
        print(f"[lcode] {n}: max|d|={d.max():.3e}")
        assert d.max() <= TOL[n], (n, d.max())
    # training this family is refused loudly, not silently mis-computed
    lat = c["latent"].to(gpu).requires_grad_(True)
    with pytest.raises(NotImplementedError):
        nerf.run_one_iter_of_nerf(512, 512, None, mc, mf, c["ro"].to(gpu), c["rd"].to(gpu), opt, mode="train", encode_position_fn=ex,
                                  encode_direction_fn=ed, expressions=c["expr"].to(gpu), background_prior=c["bg"].to(gpu),
                                  latent_code=lat)


def test_lcode_mlp_vs_fp64_oracle(hip_lib, gpu):
    import nerf
    c = C.build_case("eval_det_64_128")
    g = torch.Generator().manual_seed(2)
    ro, rd, _, _, _ = C.ray_subset(512, 512, 3, 6, 9)
    z = torch.sort(torch.rand((6, 70), generator=g) * 0.6 + 0.2, dim=-1)[0]
    p = O.init_lcode_params(6)
    m = lmodel(nerf, p, gpu)
    raw, _ = m.hip_forward(ro.to(gpu), rd.to(gpu), z.to(gpu), None, c["expr"].to(gpu), c["latent"].to(gpu), O.NEAR, O.FAR, False)
    p64 = {k: v.double() for k, v in p.items()}
    ref = O.lcode_mlp(p64, O.encode_points(ro.double(), rd.double(), z.double(), O.NEAR, O.FAR), c["expr"].double(),
                      c["latent"].double()).reshape(6, 70, 4)
    err = (raw.cpu().double() - ref).abs().amax(dim=(0, 1))
    scale = ref.abs().amax(dim=(0, 1))
    print("lcode mlp err", err.tolist(), "scale", scale.tolist())
    assert torch.all(err <= 2e-5 * scale + 2e-5)
