"""Build libnerface_hip.so (gfx950) in-tree with hipcc.  Usage: python 4d-facial-avatars_amd/build.py [--force]

The library is plain C ABI (include/nerface_hip.h); it links only against the HIP runtime, which the
host process (PyTorch-ROCm) has already loaded when the Python binding dlopens it.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "lib")
OUT = os.path.join(OUT_DIR, "libnerface_hip.so")
SOURCES = ["nf_lib.hip", "nf_rays.hip", "nf_choice.hip", "nf_render.hip", "nf_mlp.hip", "nf_mlp_bwd.hip", "nf_mlp_bf16.hip", "nf_mlp_f16.hip", "nf_mlp_f16x2.hip", "nf_mlp_f16_train.hip", "nf_mlp_f16_bwd.hip", "nf_mlp_f16_dw.hip", "nf_mlp_bf16_train.hip", "nf_mlp_bf16_bwd.hip", "nf_mlp_bf16_dw.hip", "nf_tiny.hip", "nf_flex.hip", "nf_mlp_lcode.hip", "nf_mlp_lcode_bwd.hip", "nf_mlp_lcode_bf16.hip", "nf_mlp_lcode_f16.hip", "nf_mlp_lcode_f16x2.hip", "nf_mlp_lcode_f16_train.hip", "nf_mlp_lcode_f16_bwd.hip", "nf_mlp_lcode_bf16_train.hip", "nf_mlp_lcode_bf16_bwd.hip", "nf_pipeline.hip", "nf_mlp_encoded.hip", "nf_optim.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=default",
         "-Wno-unused-result"]


# per-unit flags.  The split-fp16 forward kernels form (hi, lo) operand pairs with v_fma_mixlo / mixhi_f16 (nf_mlp_bf16_machinery.inc:
# nfb_to_operands_f16); the SLP vectoriser would pair the scalar f32 operations of that epilogue into v_pk_mul / v_pk_fma_f32 -- which
# defeats the mix patterns and costs more issue time beside MFMAs than the scalar forms (MI355X_MICROARCH.md, "price of one filler").
UNIT_FLAGS = {u: ["-fno-slp-vectorize"] for u in ("nf_mlp_f16.hip", "nf_mlp_f16x2.hip", "nf_mlp_f16_train.hip", "nf_mlp_lcode_f16.hip",
                                                  "nf_mlp_lcode_f16x2.hip", "nf_mlp_lcode_f16_train.hip")}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build libnerface_hip.so)")


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(SRC, f) for f in os.listdir(SRC)] + [os.path.join(HERE, "..", "include", "nerface_hip.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def _includes(src: str, seen=None) -> set:
    """Transitive `#include "..."` closure of one source (so that an edit recompiles only the units that see it)."""
    import re
    seen = set() if seen is None else seen
    if src in seen or not os.path.exists(src):
        return seen
    seen.add(src)
    for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', open(src).read(), re.M):
        _includes(os.path.normpath(os.path.join(os.path.dirname(src), m.group(1))), seen)
    return seen


def build_variant(suffix: str, extra_flags, verbose: bool = True) -> str:
    """Experiment builds: lib/libnerface_hip_<suffix>.so compiled with extra -D switches (own object directory).  Selected at
    run time with NERFACE_HIP_LIB=<path>; never loaded by default.  Used for same-session A/B kernel timings (profiles/)."""
    global OUT
    keep = OUT
    OUT = os.path.join(OUT_DIR, f"libnerface_hip_{suffix}.so")
    try:
        return build(force=False, verbose=verbose, _extra=list(extra_flags), _obj=f"obj_{suffix}", _always=True)
    finally:
        OUT = keep


def build(force: bool = False, verbose: bool = True, _extra=(), _obj="obj", _always=False) -> str:
    if not force and not _always and not _stale():
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    obj_dir = os.path.join(OUT_DIR, _obj)                       # object cache (git-ignored): incremental rebuilds
    os.makedirs(obj_dir, exist_ok=True)
    objs = []
    cc = _hipcc()
    procs = []
    for s in SOURCES:
        src = os.path.join(SRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
        objs.append(obj)
        newest = max(os.path.getmtime(d) for d in _includes(src) | {os.path.abspath(__file__)})
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > newest:
            continue
        # -Rpass-analysis: per-kernel registers / spills / scratch / LDS as compiler remarks, kept beside the object
        # (lib/obj/<unit>.usage.txt; tests/test_host.py reads them: no kernel of the library may spill)
        procs.append((s, subprocess.Popen([cc, *FLAGS, *UNIT_FLAGS.get(s, []), *_extra, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        try:
            out, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            p.kill()
            raise RuntimeError(f"hipcc timed out on {s}")
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out.decode()}")
        text = out.decode()
        usage = [ln for ln in text.splitlines() if "-Rpass-analysis=kernel-resource-usage" in ln]
        with open(os.path.join(obj_dir, s.replace(".hip", ".usage.txt")), "w") as f:
            f.write("\n".join(usage) + "\n")
        rest = "\n".join(ln for ln in text.splitlines() if "kernel-resource-usage" not in ln and not ln.lstrip().startswith(("|", "^", "In file included from"))
                         and not ln.strip()[:1].isdigit())
        if verbose and rest.strip():
            print(rest)
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode())
    if verbose:
        print("built", OUT)
    return OUT


def build_tools(verbose: bool = True):
    """The stand-alone HIP probes bench.py runs as subprocesses (tools/micro/store_bw: the training kernels' store pattern with
    nothing else in the way).  Binaries stay beside their source (git-ignored; they travel with the push like the library)."""
    micro = os.path.normpath(os.path.join(HERE, "..", "tools", "micro"))
    built = []
    for name in ("store_bw",):
        src, exe = os.path.join(micro, name + ".hip"), os.path.join(micro, name)
        if os.path.exists(exe) and os.path.getmtime(exe) > os.path.getmtime(src):
            continue
        r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O2", "-Wno-unused-result", "-o", exe, src], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n" + r.stdout.decode())
        built.append(exe)
        if verbose:
            print("built", exe)
    return built


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":          # python build.py --variant NAME -DFOO=1 ...
        print(build_variant(sys.argv[2], sys.argv[3:]))
    else:
        build(force="--force" in sys.argv)
        build_tools()
