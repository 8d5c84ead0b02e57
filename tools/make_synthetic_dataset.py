"""Write a tiny synthetic NeRFace dataset in the on-disk format load_flame_data reads (transforms_*.json, PNG frames,
bg/00050.png, index_map.npy).  Used by the launcher tests; also handy for smoke-testing the drop-in scripts."""
import json
import math
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def frame_pose(f):
    """Camera-to-world matrix of synthetic frame f: a small yaw / pitch orbit 0.5 in front of the head."""
    import math
    a = 0.3 * math.sin(2 * math.pi * f / 100.0)
    b = 0.15 * math.cos(2 * math.pi * f / 100.0)
    ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
    m = np.eye(4)
    m[:3, :3] = ry @ rx
    m[:3, 3] = [0.02 * math.sin(2 * math.pi * f / 100.0), 0.02 * math.cos(2 * math.pi * f / 100.0), 0.5]
    return m

def write(basedir, size=32, n_train=6, n_val=2, n_test=3, seed=0):
    rng = np.random.RandomState(seed)
    os.makedirs(os.path.join(basedir, "bg"), exist_ok=True)
    bg = (rng.rand(size, size, 3) * 255).astype(np.uint8)
    Image.fromarray(bg).save(os.path.join(basedir, "bg", "00050.png"))
    f = 0
    index_map = []
    for split, n in (("train", n_train), ("val", n_val), ("test", n_test)):
        os.makedirs(os.path.join(basedir, split), exist_ok=True)
        frames = []
        for k in range(n):
            img = (0.5 * bg + 0.5 * rng.rand(size, size, 3) * 255).astype(np.uint8)
            name = f"{split}/f_{k:04d}"
            Image.fromarray(img).save(os.path.join(basedir, name + ".png"))
            frames.append({"file_path": name, "bbox": [0.25, 0.75, 0.25, 0.75], "transform_matrix": frame_pose(f).tolist(),
                           "expression": (0.5 * rng.randn(76)).tolist()})
            if split == "test":
                index_map.append([k, k % n_train])
            f += 1
        meta = {"camera_angle_x": 2 * math.atan(0.5 * size / (1.5 * size)), "intrinsics": [-1.5 * size, 1.5 * size, 0.5, 0.5],
                "frames": frames}
        with open(os.path.join(basedir, f"transforms_{split}.json"), "w") as fp:
            json.dump(meta, fp)
    np.save(os.path.join(basedir, "index_map.npy"), np.array(index_map))
    return basedir


def config(basedir, logdir, train_iters=6, num_random_rays=256, model_type="ConditionalBlendshapePaperNeRFModel"):
    model = dict(type=model_type, num_layers=4, hidden_size=256, skip_connect_every=3, include_input_xyz=True,
                 log_sampling_xyz=True, num_encoding_fn_xyz=10, use_viewdirs=True, include_input_dir=False, num_encoding_fn_dir=4,
                 log_sampling_dir=True)
    mode = dict(num_random_rays=num_random_rays, chunksize=2048, perturb=True, num_coarse=64, num_fine=64, white_background=False,
                radiance_field_noise_std=0.1, lindisp=False)
    val = dict(mode)
    val.update(chunksize=65536, radiance_field_noise_std=0.0)
    return dict(experiment=dict(id="synthetic", logdir=logdir, randomseed=42, train_iters=train_iters, validate_every=1000, save_every=5,
                                print_every=2, device=0),
                dataset=dict(type="blender", basedir=basedir, half_res=False, testskip=1, no_ndc=True, near=0.2, far=0.8),
                models=dict(coarse=dict(model), fine=dict(model)),
                optimizer=dict(type="Adam", lr=5.0e-4), scheduler=dict(lr_decay=250, lr_decay_factor=0.1),
                nerf=dict(use_viewdirs=True, encode_position_fn="positional_encoding", encode_direction_fn="positional_encoding",
                          train=mode, validation=val))


if __name__ == "__main__":
    import yaml
    base = sys.argv[1]
    write(os.path.join(base, "data"))
    with open(os.path.join(base, "config.yml"), "w") as f:
        yaml.safe_dump(config(os.path.join(base, "data"), os.path.join(base, "logs")), f)
    print("wrote", base)
