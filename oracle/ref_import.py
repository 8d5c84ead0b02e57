"""Import the UNMODIFIED reference `nerf` package: from /root/reference in the build container, from the archive
oracle/_ref/nerface_ref.zip (packed by oracle/make_ref.py from that tree, byte for byte; git-ignored, ships with the push)
on the GPU box.

TEST INFRASTRUCTURE ONLY.  ``reference_available()`` = the live tree is there (build container);
``reference_importable()`` = live tree OR the travelling archive.  Five third-party modules that the reference imports
at module scope but never calls on the hot path are stubbed; no reference source is modified, and
none enters the repository history.
"""
from __future__ import annotations

import contextlib
import importlib
import os
import sys
import types

REF_ROOT = "/root/reference/nerface_code/nerf-pytorch"
REF_ARCHIVE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "nerface_ref.zip")


def reference_available() -> bool:
    """The live, unpacked reference tree (build container only: configs, command files, everything)."""
    return os.path.isdir(os.path.join(REF_ROOT, "nerf"))


def reference_importable() -> bool:
    """The reference's `nerf` package and its two scripts can be imported: live tree, or the archive that travelled."""
    return reference_available() or os.path.exists(REF_ARCHIVE)


def import_root() -> str:
    """sys.path entry the unmodified reference modules are imported from (a directory, or the zip: zipimport)."""
    if reference_available():
        return REF_ROOT
    if os.path.exists(REF_ARCHIVE):
        return REF_ARCHIVE
    raise RuntimeError("neither /root/reference nor oracle/_ref/nerface_ref.zip is present (python -m oracle.make_ref packs it "
                       "in the build container)")


def reference_kind() -> str:
    return "live tree /root/reference" if reference_available() else "oracle/_ref/nerface_ref.zip (unmodified files packed by oracle/make_ref.py)"


def import_reference():
    """Returns the reference's `nerf` package as a module object registered under the private name
    ``_ref_nerf`` so it cannot shadow (or be shadowed by) the product package that is also called
    ``nerf``."""
    if "_ref_nerf" in sys.modules:
        return sys.modules["_ref_nerf"]
    root = import_root()
    for n in ["pytorch3d", "pytorch3d.transforms", "torchsearchsorted", "cv2", "imageio"]:
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["pytorch3d"].transforms = sys.modules["pytorch3d.transforms"]
    saved = {k: v for k, v in sys.modules.items() if k == "nerf" or k.startswith("nerf.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, root)
    try:
        mod = importlib.import_module("nerf")
    finally:
        sys.path.remove(root)
    # re-home: reference modules live on as _ref_nerf*, the public name is released again
    for k in [k for k in sys.modules if k == "nerf" or k.startswith("nerf.")]:
        sys.modules["_ref_" + k] = sys.modules.pop(k)
    sys.modules.update(saved)
    return mod


def import_reference_tiny():
    """The reference's tiny_nerf.py script module (BASELINE config 1), imported as ``_ref_tiny_nerf``
    with the reference `nerf` package temporarily visible under its public name."""
    if "_ref_tiny_nerf" in sys.modules:
        return sys.modules["_ref_tiny_nerf"]
    saved = _expose_reference_nerf()
    root = import_root()
    mine = sys.modules.pop("tiny_nerf", None)          # the product has a module of the same name (bench.py imports it first)
    sys.path.insert(0, root)
    try:
        import matplotlib
        matplotlib.use("Agg")
        mod = importlib.import_module("tiny_nerf")
        assert getattr(mod, "__file__", "").startswith(root), (mod.__file__, root)
        sys.modules["_ref_tiny_nerf"] = sys.modules.pop("tiny_nerf")
    finally:
        sys.path.remove(root)
        _hide_reference_nerf(saved)
        if mine is not None:
            sys.modules["tiny_nerf"] = mine
    return mod


def _expose_reference_nerf():
    """Make the reference `nerf` package importable under its public name; returns what to restore afterwards."""
    import_reference()
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "nerf" or k.startswith("nerf.")}
    for k in [k for k in sys.modules if k.startswith("_ref_nerf")]:
        sys.modules[k[len("_ref_"):]] = sys.modules[k]
    return saved


def _hide_reference_nerf(saved):
    for k in [k for k in sys.modules if k == "nerf" or k.startswith("nerf.")]:
        del sys.modules[k]
    sys.modules.update(saved)


def _torchvision_stub():
    """torchvision is not installed here.  The eval script uses exactly one thing from it, transforms.ToPILImage() on a
    float (3, H, W) tensor (EV:184-190); torchvision (pinned 0.6.0 in nerf.yml:75) documents and implements that as
    `pic.mul(255).byte()` -> HWC numpy -> PIL.Image.fromarray.  The stub restates just that."""
    import numpy as np
    from PIL import Image
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")

    class ToPILImage:
        def __call__(self, pic):
            if pic.is_floating_point():
                pic = pic.mul(255).byte()
            return Image.fromarray(np.transpose(pic.cpu().numpy(), (1, 2, 0)))

    tr.ToPILImage = ToPILImage
    tv.transforms = tr
    return tv, tr


def import_reference_eval():
    """The reference's eval_transformed_rays.py script module (for torch_normal_map EV:84-119 and cast_to_image EV:184-190),
    imported unmodified as ``_ref_eval`` (its `main()` sits behind `if __name__ == "__main__"`).  torchvision and imageio
    are stubbed (see _torchvision_stub); matplotlib/tqdm/yaml are real."""
    if "_ref_eval" in sys.modules:
        return sys.modules["_ref_eval"]
    saved = _expose_reference_nerf()
    added = []
    if "torchvision" not in sys.modules:
        tv, tr = _torchvision_stub()
        sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tr
        added += ["torchvision", "torchvision.transforms"]
    root = import_root()
    sys.path.insert(0, root)
    try:
        import matplotlib
        matplotlib.use("Agg")
        mod = importlib.import_module("eval_transformed_rays")
    finally:
        sys.path.remove(root)
        _hide_reference_nerf(saved)
        for k in added:
            sys.modules.pop(k, None)
    sys.modules["_ref_eval"] = sys.modules.pop("eval_transformed_rays")
    return mod


def cv2_area_resize(img, dsize):
    """OpenCV is not installed here.  What load_flame.py:171-175 needs is cv2.resize(float32 HxWx3, dsize=(w, h),
    interpolation=INTER_AREA) with an integer shrink factor: OpenCV (pinned opencv-python-headless 4.2.0.34, nerf.yml:93)
    takes its resizeAreaFast_ path for that -- per output element a float accumulator summed over the source block in
    row-major order, then `sum * (1/area)`.  3-channel float images do not take the SIMD shortcut (ResizeAreaFastVec_SIMD_32f
    handles cn 1 and 4 only), so the scalar order is the order.  Restated here for the reference's `cv2` stub only."""
    import numpy as np
    h, w = img.shape[:2]
    ow, oh = int(dsize[0]), int(dsize[1])
    fy, fx = h // oh, w // ow
    assert fy * oh == h and fx * ow == w, "restated for integer shrink factors only"
    acc = np.zeros((oh, ow) + img.shape[2:], dtype=np.float32)
    for sy in range(fy):
        for sx in range(fx):
            acc = (acc + img[sy::fy, sx::fx][:oh, :ow]).astype(np.float32)
    return (acc * np.float32(1.0 / (fx * fy))).astype(np.float32)


@contextlib.contextmanager
def flame_loader_io():
    """For the duration of the block the reference's nerf.load_flame sees an `imageio.imread` backed by PIL (imageio's own
    PNG reader is Pillow) and a `cv2.resize` / `cv2.INTER_AREA` backed by cv2_area_resize.  Module attributes of the
    stub modules are set and removed again; no reference file is touched."""
    import numpy as np
    from PIL import Image
    import_reference()
    lf = sys.modules["_ref_nerf.load_flame"]

    def imread(path):
        with Image.open(path) as im:
            return np.asarray(im)

    lf.imageio.imread = imread
    lf.cv2.INTER_AREA = 3
    lf.cv2.resize = lambda img, dsize=None, interpolation=None: cv2_area_resize(img, dsize)
    try:
        yield lf
    finally:
        del lf.imageio.imread, lf.cv2.INTER_AREA, lf.cv2.resize


@contextlib.contextmanager
def relu_clone_shim(ref):
    """Quirk Q9: on torch>=1.10 the reference's in-place ``sigma_a[:, -1] += 1e-6`` on a ReLU output
    breaks autograd.  For the duration of the block, volume_render_radiance_field sees a ReLU that
    returns a fresh tensor; the reference files are untouched and gradient semantics are unchanged."""
    import torch
    tu = sys.modules["_ref_nerf.train_utils"]
    orig = tu.volume_render_radiance_field

    def wrapped(*a, **k):
        f = torch.nn.functional
        keep = f.relu
        f.relu = lambda x, *aa, **kk: torch.relu(x).clone()
        try:
            return orig(*a, **k)
        finally:
            f.relu = keep

    tu.volume_render_radiance_field = wrapped
    try:
        yield
    finally:
        tu.volume_render_radiance_field = orig


@contextlib.contextmanager
def injected_random(rand_list, randn_list):
    """Serve torch.rand / torch.randn calls from pre-generated tensors (parity mode)."""
    import torch
    r_it, n_it = iter(rand_list), iter(randn_list)
    keep_r, keep_n = torch.rand, torch.randn

    def fake_rand(*shape, **kw):
        t = next(r_it)
        want = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        assert tuple(t.shape) == want, (t.shape, want)
        return t.clone()

    def fake_randn(*shape, **kw):
        t = next(n_it)
        want = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else tuple(shape)
        assert tuple(t.shape) == want, (t.shape, want)
        return t.clone()

    torch.rand, torch.randn = fake_rand, fake_randn
    try:
        yield
    finally:
        torch.rand, torch.randn = keep_r, keep_n
