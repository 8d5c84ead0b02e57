"""Import the UNMODIFIED reference `nerf` package from /root/reference (build container only).

TEST INFRASTRUCTURE ONLY.  /root/reference does not exist on the GPU box; everything that calls this
module guards on ``reference_available()``.  Five third-party modules that the reference imports at
module scope but never calls on the hot path are stubbed; no reference source is
modified or copied.
"""
from __future__ import annotations

import contextlib
import importlib
import os
import sys
import types

REF_ROOT = "/root/reference/nerface_code/nerf-pytorch"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "nerf"))


def import_reference():
    """Returns the reference's `nerf` package as a module object registered under the private name
    ``_ref_nerf`` so it cannot shadow (or be shadowed by) the product package that is also called
    ``nerf``."""
    if "_ref_nerf" in sys.modules:
        return sys.modules["_ref_nerf"]
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only inside the build container)")
    for n in ["pytorch3d", "pytorch3d.transforms", "torchsearchsorted", "cv2", "imageio"]:
        if n not in sys.modules:
            sys.modules[n] = types.ModuleType(n)
    sys.modules["pytorch3d"].transforms = sys.modules["pytorch3d.transforms"]
    saved = {k: v for k, v in sys.modules.items() if k == "nerf" or k.startswith("nerf.")}
    for k in saved:
        del sys.modules[k]
    sys.path.insert(0, REF_ROOT)
    try:
        mod = importlib.import_module("nerf")
    finally:
        sys.path.remove(REF_ROOT)
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
    import_reference()
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "nerf" or k.startswith("nerf.")}
    for k in [k for k in sys.modules if k.startswith("_ref_nerf")]:
        sys.modules[k[len("_ref_"):]] = sys.modules[k]
    sys.path.insert(0, REF_ROOT)
    try:
        import matplotlib
        matplotlib.use("Agg")
        mod = importlib.import_module("tiny_nerf")
    finally:
        sys.path.remove(REF_ROOT)
        for k in [k for k in sys.modules if k == "nerf" or k.startswith("nerf.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    sys.modules["_ref_tiny_nerf"] = sys.modules.pop("tiny_nerf")
    return mod


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
