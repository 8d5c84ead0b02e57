"""Execute the reference's UNMODIFIED scripts (train_transformed_rays.py TR:24-575, eval_transformed_rays.py EV:201-498).
TEST INFRASTRUCTURE ONLY (used by tests/test_gpu_dropin_scripts.py and oracle/make_golden.py).

The script files are imported byte for byte from the live tree or from oracle/_ref/nerface_ref.zip (oracle/make_ref.py);
`against="product"` resolves their `from nerf import ...` to the MI355X package (4d-facial-avatars_amd first on sys.path: the whole
drop-in switch), `against="reference"` to the reference's own `nerf` (golden generation on the CPU of the build container).

Only what this image lacks is stubbed, for the duration of the call:
  torchvision              -> transforms.ToPILImage restated (ref_import._torchvision_stub; used by cast_to_image TR:575, EV:184-190)
  torch.utils.tensorboard  -> a SummaryWriter that records add_scalar / add_image calls (the `tensorboard` package is absent)
  imageio                  -> imwrite / imread backed by PIL (imageio's own PNG plugin is Pillow)
  cv2                      -> resize(INTER_AREA) restated (ref_import.cv2_area_resize; only the reference's loader calls it)
One module attribute of the eval script is replaced when `max_frames` is given: its `tqdm` (EV:392 `for i, expression in
enumerate(tqdm(render_expressions))`), by a pass-through that stops after `max_frames` frames -- the shipped loop reads pose
240 + i for EVERY test frame (EV:433), so it cannot run to the end of any sequence anyway.
"""
from __future__ import annotations

import contextlib
import importlib
import itertools
import os
import sys
import types

from . import ref_import as RI

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "4d-facial-avatars_amd")


class RecordingWriter:
    """Stands in for torch.utils.tensorboard.SummaryWriter: keeps what the script logs."""
    last = None

    def __init__(self, logdir=None, *a, **k):
        self.logdir, self.scalars, self.images = logdir, [], []
        RecordingWriter.last = self

    def add_scalar(self, tag, value, step=None, *a, **k):
        self.scalars.append((tag, float(value), step))

    def add_image(self, tag, img, step=None, *a, **k):
        self.images.append((tag, tuple(getattr(img, "shape", ())), step))

    def flush(self):
        pass

    def close(self):
        pass


@contextlib.contextmanager
def script_stubs(written=None):
    """Install the stubs described in the module docstring; `written` (dict) receives {path: uint8 array} of every imageio.imwrite."""
    import numpy as np
    from PIL import Image
    added = []

    def put(name, mod):
        if name not in sys.modules:
            sys.modules[name] = mod
            added.append(name)
            return mod
        return sys.modules[name]

    tv, tr = RI._torchvision_stub()
    put("torchvision", tv)
    put("torchvision.transforms", tr)
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = RecordingWriter
    put("torch.utils.tensorboard", tb)
    io = put("imageio", types.ModuleType("imageio"))
    keep_io = {k: getattr(io, k, None) for k in ("imwrite", "imread")}

    def imwrite(path, arr, *a, **k):
        arr = np.asarray(arr)
        Image.fromarray(arr).save(path)
        if written is not None:
            written[str(path)] = arr.copy()

    def imread(path, *a, **k):
        with Image.open(path) as im:
            return np.asarray(im)

    io.imwrite, io.imread = imwrite, imread
    cv = put("cv2", types.ModuleType("cv2"))
    keep_cv = {k: getattr(cv, k, None) for k in ("resize", "INTER_AREA")}
    if not hasattr(cv, "resize"):
        cv.INTER_AREA = 3
        cv.resize = lambda img, dsize=None, interpolation=None: RI.cv2_area_resize(img, dsize)
    try:
        yield
    finally:
        for k, v in keep_io.items():
            if v is None:
                if hasattr(io, k):
                    delattr(io, k)
            else:
                setattr(io, k, v)
        for k, v in keep_cv.items():
            if v is None and hasattr(cv, k):
                delattr(cv, k)
        for n in added:
            sys.modules.pop(n, None)


def import_script(name: str, against: str):
    """Fresh import of the unmodified script `name` (no .py) with `nerf` resolved to the product or to the reference package.
    Returns the module, registered as _dropin_<name> / _refrun_<name>; must be called inside script_stubs()."""
    assert against in ("product", "reference")
    alias = ("_dropin_" if against == "product" else "_refrun_") + name
    if alias in sys.modules:
        return sys.modules[alias]
    import matplotlib
    matplotlib.use("Agg")
    root = RI.import_root()
    saved = None
    if against == "reference":
        saved = RI._expose_reference_nerf()
        path_add = [root]
    else:
        for k in [k for k in sys.modules if k == "nerf" or k.startswith("nerf.")]:
            f = getattr(sys.modules[k], "__file__", "") or ""
            if not f.startswith(PKG):
                raise RuntimeError(f"sys.modules[{k!r}] is not the product package ({f})")
        path_add = [PKG, root]                                    # the product FIRST: that is the whole drop-in switch
    old_path = list(sys.path)
    sys.path[:0] = path_add
    try:
        sys.modules.pop(name, None)
        mod = importlib.import_module(name)
    finally:
        sys.path[:] = old_path
        if saved is not None:
            RI._hide_reference_nerf(saved)
    sys.modules[alias] = sys.modules.pop(name)
    return mod


def run_main(mod, argv, max_frames=None):
    """mod.main() with sys.argv = argv (the scripts parse sys.argv).  max_frames: see the module docstring (eval script only)."""
    keep_argv = sys.argv
    keep_tqdm = getattr(mod, "tqdm", None)
    if max_frames is not None and keep_tqdm is not None:
        def limited(it, *a, **k):
            return itertools.islice(it, max_frames)
        limited.write = keep_tqdm.write
        mod.tqdm = limited
    sys.argv = [getattr(mod, "__file__", "script")] + list(argv)
    try:
        return mod.main()
    finally:
        sys.argv = keep_argv
        if keep_tqdm is not None:
            mod.tqdm = keep_tqdm


def synthetic_checkpoint(path, n_train, size, seed=0):
    """A checkpoint dictionary with the schema of TR:555-568 holding seeded weights (oracle.init_paper_params: the "hard" density head
    of the golden cases), a seeded latent table and no optimizer state: the input of the as-shipped eval fixture."""
    import torch
    from . import nerface_oracle as O
    g = torch.Generator().manual_seed(100 + seed)
    ck = {"iter": 0, "model_coarse_state_dict": O.init_paper_params(2 * seed), "model_fine_state_dict": O.init_paper_params(2 * seed + 1),
          "optimizer_state_dict": {}, "loss": torch.tensor(0.0), "psnr": 0.0, "background": torch.rand((size, size, 3), generator=g),
          "latent_codes": 0.1 * torch.randn((n_train, 32), generator=g)}
    torch.save(ck, path)
    return ck


def as_shipped_case(base, size=32, n_train=6, n_test=243, seed=0):
    """The synthetic on-disk case of the as-shipped eval fixture: dataset (tools/make_synthetic_dataset.py; 243 test frames because
    the shipped loop reads pose 240 + i, EV:433), YAML config with deterministic validation sampling, seeded checkpoint.
    Returns (config path, checkpoint path)."""
    import yaml
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import make_synthetic_dataset as MS
    finally:
        sys.path.remove(os.path.join(ROOT, "tools"))
    data = os.path.join(base, "data")
    MS.write(data, size=size, n_train=n_train, n_val=2, n_test=n_test, seed=seed)
    cfg = MS.config(data, os.path.join(base, "logs"))
    cfg["nerf"]["validation"].update(perturb=False, num_coarse=64, num_fine=128)
    cfg_path = os.path.join(base, "config.yml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    ck_path = os.path.join(base, "seeded.ckpt")
    synthetic_checkpoint(ck_path, n_train, size, seed)
    return cfg_path, ck_path
