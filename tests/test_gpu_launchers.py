"""GPU: the two launchers end to end on a synthetic on-disk dataset (reference formats: transforms_*.json, PNGs,
bg/00050.png, index_map.npy, YAML config, checkpoint dictionary)."""
import os
import sys

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_then_eval_roundtrip(hip_lib, gpu, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))
    import make_synthetic_dataset as MS
    from launch import eval_sharded, train_sharded
    base = str(tmp_path)
    MS.write(os.path.join(base, "data"))
    cfg_path = os.path.join(base, "config.yml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(MS.config(os.path.join(base, "data"), os.path.join(base, "logs")), f)
    logdir = train_sharded.main(["--config", cfg_path])
    ck_path = os.path.join(logdir, "checkpoint00005.ckpt")
    assert os.path.exists(os.path.join(logdir, "checkpoint00000.ckpt")) and os.path.exists(ck_path)
    ck = torch.load(ck_path, map_location="cpu")
    assert set(ck) == {"iter", "model_coarse_state_dict", "model_fine_state_dict", "optimizer_state_dict", "loss", "psnr",
                       "background", "latent_codes"}
    assert ck["latent_codes"].shape == (6, 32) and len(ck["optimizer_state_dict"]["param_groups"]) == 2
    assert float(ck["latent_codes"].abs().sum()) > 0                     # latent rows did train
    # resume runs and keeps training the SAME latent tensor
    cfg = yaml.safe_load(open(cfg_path))
    cfg["experiment"]["train_iters"] = 8
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    train_sharded.main(["--config", cfg_path, "--load-checkpoint", ck_path])
    ck2 = torch.load(os.path.join(logdir, "checkpoint00007.ckpt"), map_location="cpu")
    assert not torch.equal(ck2["latent_codes"], ck["latent_codes"])
    # eval: 3 test frames -> PNGs, identical for both precisions to within quantisation
    out = os.path.join(base, "render")
    frames = eval_sharded.main(["--config", cfg_path, "--checkpoint", ck_path, "--savedir", out, "--save-disparity-image",
                                "--save-normals", "--precision", "f32"])
    assert np.asarray(__import__("PIL.Image").Image.open(os.path.join(out, "normals", "0000.png"))).shape == (31, 31, 3)
    assert frames == [0, 1, 2]
    from PIL import Image
    a = np.asarray(Image.open(os.path.join(out, "0001.png")))
    assert a.shape == (32, 32, 3) and a.std() > 0
    out2 = os.path.join(base, "render_b")
    eval_sharded.main(["--config", cfg_path, "--checkpoint", ck_path, "--savedir", out2, "--precision", "bf16x3"])
    b = np.asarray(Image.open(os.path.join(out2, "0001.png")))
    assert a.shape == b.shape                      # (perturb=True in validation, as shipped: images differ by sampling noise)
    assert os.path.exists(os.path.join(out, "disparity", "0002.png"))


def test_launchers_second_model_family(hip_lib, gpu, tmp_path):
    """The same two launchers with `type: ConditionalBlendshapeLearnableCodeNeRFModel` in the config (as 6 shipped configs
    have): trains (exact-f32 kernels), checkpoints with this family's state_dict keys, renders in both precisions."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))
    import make_synthetic_dataset as MS
    from launch import eval_sharded, train_sharded
    from oracle import nerface_oracle as O
    from PIL import Image
    base = str(tmp_path)
    MS.write(os.path.join(base, "data"))
    cfg_path = os.path.join(base, "config.yml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(MS.config(os.path.join(base, "data"), os.path.join(base, "logs"),
                                 model_type="ConditionalBlendshapeLearnableCodeNeRFModel"), f)
    logdir = train_sharded.main(["--config", cfg_path])
    ck0 = torch.load(os.path.join(logdir, "checkpoint00000.ckpt"), map_location="cpu")
    ck = torch.load(os.path.join(logdir, "checkpoint00005.ckpt"), map_location="cpu")
    assert list(ck["model_fine_state_dict"].keys()) == O.LCODE_KEYS
    moved = [k for k in O.LCODE_KEYS if not torch.equal(ck["model_fine_state_dict"][k], ck0["model_fine_state_dict"][k])]
    assert len(moved) == len(O.LCODE_KEYS), set(O.LCODE_KEYS) - set(moved)        # every tensor of the family receives gradients
    assert float(ck["latent_codes"].abs().sum()) > 0 and np.isfinite(float(ck["loss"]))
    for prec in ("f32", "bf16x3"):
        out = os.path.join(base, "render_" + prec)
        assert eval_sharded.main(["--config", cfg_path, "--checkpoint", os.path.join(logdir, "checkpoint00005.ckpt"), "--savedir", out,
                                  "--precision", prec]) == [0, 1, 2]
        a = np.asarray(Image.open(os.path.join(out, "0001.png")))
        assert a.shape == (32, 32, 3) and a.std() > 0


def test_eval_postprocess_matches_oracle(hip_lib, gpu):
    from nerf import ops
    from oracle import nerface_oracle as O
    g = torch.Generator().manual_seed(8)
    rgb = torch.rand((48, 48, 3), generator=g) * 1.4 - 0.2
    disp = torch.rand((48, 48), generator=g) * 0.5 + 1.0
    w = torch.rand((48, 48), generator=g) * 0.5
    u8, nrm = ops.eval_postprocess(rgb.to(gpu), disp.to(gpu), w.to(gpu), O.INTRINSICS)
    assert torch.equal(u8.cpu(), O.cast_to_u8(rgb))
    want = O.normal_map(disp, O.INTRINSICS, w)
    d = (nrm.cpu().int() - want.int()).abs()
    assert nrm.shape == (47, 47, 3) and int(d.max()) <= 1 and float((d == 0).float().mean()) > 0.99     # truncation at an ulp boundary
