"""Tie bench.py's cpu_baseline (kind "port": oracle/nerface_oracle.py) to the reference's own CPU path.  BUILD CONTAINER ONLY
(/root/reference must exist).  TEST INFRASTRUCTURE.

Both run the same rays of the same synthetic frame (eval, 64+128, deterministic sampling) with the same torch thread count in
this process: the UNMODIFIED reference `run_one_iter_of_nerf` (stub-imported, oracle/ref_import.py) and the oracle port.
Writes profiles/r02_port_vs_reference_cpu.json; bench.py copies it into `cpu_baseline.reference_ratio` (the reference
cannot travel to the GPU box).   python -m oracle.time_port_vs_reference [n_rays]"""
from __future__ import annotations

import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cases as C                      # noqa: E402
from oracle import make_golden as MG               # noqa: E402
from oracle import ref_import as RI                # noqa: E402


def main():
    n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    ref = RI.import_reference()
    threads = min(os.cpu_count() or 1, 8)
    torch.set_num_threads(threads)
    c = C.build_case("eval_det_64_128")
    ro, rd, bg, tgt, idx = C.ray_subset(512, 512, 3, n_rays, seed=5)
    c.update(n_rays=n_rays, ro=ro, rd=rd, bg=bg, tgt=tgt, idx=idx)
    warm = dict(c)
    warm.update(ro=ro[:128], rd=rd[:128], bg=bg[:128])
    res = {}
    with torch.no_grad():
        for name, fn in (("reference", lambda cc: MG.run_reference(ref, cc)[0]), ("port", lambda cc: C.run_oracle(cc))):
            fn(warm)
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                out = fn(c)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            res[name] = {"seconds": best, "rays_per_s": n_rays / best}
            res[name + "_out"] = out
    same = all(torch.equal(a, b) for a, b in zip(res.pop("reference_out"), res.pop("port_out")))
    blob = {"what": "unmodified reference run_one_iter_of_nerf vs oracle port, same process, same rays, torch-CPU fp32",
            "sample": f"{n_rays} rays of one 512x512 frame, 64+128 samples, deterministic sampling, chunksize 65536",
            "threads": threads, "host": "build container (Intel Xeon @ 2.1 GHz, 8 cores)", "torch": torch.__version__,
            "reference": res["reference"], "port": res["port"], "port_over_reference_time": res["port"]["seconds"] / res["reference"]["seconds"],
            "outputs_bit_identical": same}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_port_vs_reference_cpu.json")
    json.dump(blob, open(out, "w"), indent=1)
    print(json.dumps(blob))


if __name__ == "__main__":
    main()
