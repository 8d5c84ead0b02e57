"""BASELINE config 1 (tiny_nerf 64x64, 32 samples): the HIP tiny path against the reference's own output (golden) and the
oracle.  GPU only."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import nerface_oracle as O
from tests import util as U

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tiny_nerf_forward_matches_reference_output(hip_lib, gpu):
    sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))
    import tiny_nerf as TN
    gold = np.load(os.path.join(ROOT, "tests", "golden", "tiny_64x64x32.npz"))["rgb"]
    model = TN.VeryTinyNerfModel(num_encoding_functions=10)
    model.load_state_dict(O.tiny_init_params(9458))
    model.to(gpu)
    pose = O.frame_pose(7)
    pose[:3, 3] = torch.tensor([0.3, -0.2, 4.0])
    jit = torch.rand((64, 64, 32), generator=torch.Generator().manual_seed(77))
    with torch.no_grad(), U.injected_random([jit], []):
        rgb = TN.run_one_iter_of_tinynerf(64, 64, torch.tensor(138.88 * 64 / 100.0), pose.to(gpu), 2.0, 6.0, 32,
                                          lambda x, n: TN.positional_encoding(x, n), TN.get_minibatches, 16384, model, 10)
    d = np.abs(rgb.cpu().numpy() - gold)
    print("tiny max|d| vs reference output:", d.max())
    assert rgb.shape == (64, 64, 3) and d.max() < 5e-6
    with pytest.raises(NotImplementedError):
        TN.run_one_iter_of_tinynerf(64, 64, torch.tensor(88.9), pose.to(gpu), 2.0, 6.0, 32, None, None, 16384, model, 10)   # grad mode


def test_render_volume_density_matches_oracle(hip_lib, gpu):
    sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))
    import tiny_nerf as TN
    g = torch.Generator().manual_seed(3)
    raw = torch.randn((5, 7, 32, 4), generator=g) * 2
    depth = torch.sort(torch.rand((5, 7, 32), generator=g) * 4 + 2, dim=-1)[0]
    rgb, dmap, acc = TN.render_volume_density(raw.to(gpu), torch.zeros(5, 7, 3, device=gpu), depth.to(gpu))
    sigma = torch.relu(raw[..., 3].double())
    dists = torch.cat((depth[..., 1:] - depth[..., :-1], torch.full_like(depth[..., :1], 1e10)), -1).double()
    alpha = 1 - torch.exp(-sigma * dists)
    T = torch.cumprod(1 - alpha + 1e-10, -1)
    T = torch.cat((torch.ones_like(T[..., :1]), T[..., :-1]), -1)
    w = alpha * T
    assert ((w[..., None] * torch.sigmoid(raw[..., :3].double())).sum(-2) - rgb.cpu().double()).abs().max() < 3e-6
    assert ((w * depth.double()).sum(-1) - dmap.cpu().double()).abs().max() < 2e-5
    assert (w.sum(-1) - acc.cpu().double()).abs().max() < 3e-6
