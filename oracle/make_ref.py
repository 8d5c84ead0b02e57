"""Recipe for oracle/_ref/: pack the UNMODIFIED reference path into one archive that travels to the GPU box.  TEST INFRASTRUCTURE.

    python -m oracle.make_ref            (also run by __graft_entry__.build() whenever /root/reference is present)

/root/reference exists only in the build container.  bench.py's `cpu_baseline` leg has to time the reference's OWN
`run_one_iter_of_nerf` (T:165-290) + `get_ray_bundle` (H:68) on the GPU box's host cores, and the drop-in test has to execute the
UNMODIFIED `train_transformed_rays.py` / `eval_transformed_rays.py` against the product package there, so this recipe stores
the reference's `nerf/*.py` package and the two scripts -- byte for byte, straight from where they lie under /root/reference --
in `oracle/_ref/nerface_ref.zip`.  `oracle/_ref/` is git-ignored (nothing of the reference enters the history) and NOT
gpurun-ignored (it ships with the push like the built .so).  Python imports the modules straight out of the archive
(zipimport): `oracle/ref_import.py` puts the archive on sys.path when the live tree is absent.

Only `tests/`, `__graft_entry__.build()/smoke()` and the `cpu_baseline` leg of bench.py may touch oracle/ (tests/test_host.py
enforces it); the product never imports any of this.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference/nerface_code/nerf-pytorch"
OUT_DIR = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(OUT_DIR, "nerface_ref.zip")
MANIFEST = os.path.join(OUT_DIR, "manifest.json")
SCRIPTS = ("train_transformed_rays.py", "eval_transformed_rays.py", "tiny_nerf.py")


def _members():
    out = []
    pkg = os.path.join(REF_ROOT, "nerf")
    for f in sorted(os.listdir(pkg)):
        if f.endswith(".py"):
            out.append(("nerf/" + f, os.path.join(pkg, f)))
    for s in SCRIPTS:
        out.append((s, os.path.join(REF_ROOT, s)))
    return out


def build(verbose: bool = True) -> str | None:
    """(Re)write the archive if the live reference tree is present; returns its path, or None when there is nothing to pack
    (GPU box: the archive that travelled with the push is used as it is)."""
    if not os.path.isdir(os.path.join(REF_ROOT, "nerf")):
        return ARCHIVE if os.path.exists(ARCHIVE) else None
    members = _members()
    digest = {arc: hashlib.sha256(open(src, "rb").read()).hexdigest() for arc, src in members}
    if os.path.exists(ARCHIVE) and os.path.exists(MANIFEST):
        try:
            if json.load(open(MANIFEST)).get("sha256") == digest:
                return ARCHIVE
        except Exception:
            pass
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = ARCHIVE + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for arc, src in members:
            zi = zipfile.ZipInfo(arc, date_time=(2020, 1, 1, 0, 0, 0))        # fixed stamp: the archive is reproducible
            zi.compress_type = zipfile.ZIP_DEFLATED
            z.writestr(zi, open(src, "rb").read())
    os.replace(tmp, ARCHIVE)
    json.dump({"source": REF_ROOT, "files": [a for a, _ in members], "sha256": digest}, open(MANIFEST, "w"), indent=1)
    if verbose:
        print(f"oracle/_ref: packed {len(members)} unmodified reference files into {ARCHIVE}")
    return ARCHIVE


if __name__ == "__main__":
    p = build()
    print(p if p else "no reference tree and no archive")
    sys.exit(0 if p else 1)
