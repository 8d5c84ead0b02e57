"""north_star's parity gate, measured where it is hard: |PSNR(arithmetic, target) - PSNR(exact, target)| <= 1e-4 dB with
PSNR = -10 log10(mse) (the reference's `mse2psnr`, nerf_helpers.py:14-18, applied as in train_transformed_rays.py:355-392).

For an error e = arithmetic - exact against the residual r = exact - target,

    dPSNR ~ 4.34 * (mean(e^2) + 2 mean(e r)) / mean(r^2)   dB,

so the same rendering error moves the figure by 1 / mse(exact, target): a uniform-random target (mse ~ 0.17, PSNR ~ 8 dB -- the
target SURVEY 8(d) prescribes) is ~170x less sensitive than an image a trained model approaches to 30 dB (mse 1e-3), and the
cross term only averages out as 1 / sqrt(rays).  This module builds targets the exact frame approximates to a chosen PSNR
(`target_near`: clamp(exact + sigma randn), sigma calibrated), evaluates the gate on whole frames AND on scattered ray subsets
(`gate_cells`), and states which cells an arithmetic is expected to pass (`EXPECTED_PASS`, measured on MI355X:
profiles/r06_gate_sensitivity.md).  Pure torch arithmetic on tensors the caller rendered; used by tests/test_gpu_gate.py,
tools/frame_gate_sweep.py, bench.py's summary and launch/eval_sharded.py --verify-gate.
"""
from __future__ import annotations

import math

import torch

GATE_DB = 1e-4
TARGET_DBS = (20.0, 30.0, 40.0)            # PSNR(exact frame, target) of the realistic targets
SUBSET_RAYS = (1024, 3001)                 # ray subsets beside the whole frame (the sizes of tests/test_gpu_e2e.py's f32 gates)


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
    return float(-10.0 * torch.log10(torch.mean((a.double() - b.double()) ** 2)))


def target_near(exact: torch.Tensor, db: float, seed: int) -> torch.Tensor:
    """An image in [0, 1] that `exact` approximates to `db` dB: clamp(exact + sigma * randn, 0, 1), sigma calibrated by fixed-point
    iteration on the clamped image (seeded CPU generator: the same target on every device)."""
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn(tuple(exact.shape), generator=g, dtype=torch.float64).to(exact.device)
    ex = exact.double()
    sigma = 10.0 ** (-db / 20.0)
    tgt = (ex + sigma * noise).clamp(0.0, 1.0)
    for _ in range(6):
        got = psnr(ex, tgt)
        if abs(got - db) < 1e-3:
            break
        sigma *= 10.0 ** ((got - db) / 20.0)
        tgt = (ex + sigma * noise).clamp(0.0, 1.0)
    return tgt


def targets_for(exact: torch.Tensor, seed: int) -> dict:
    """{"random": uniform noise (SURVEY 8(d)'s target), "20dB" / "30dB" / "40dB": images the exact frame approximates that well}."""
    g = torch.Generator().manual_seed(seed)
    out = {"random": torch.rand(tuple(exact.shape), generator=g, dtype=torch.float64).to(exact.device)}
    for db in TARGET_DBS:
        out[f"{db:.0f}dB"] = target_near(exact, db, seed + 1 + int(db))
    return out


def gate_cells(exact: torch.Tensor, approx: torch.Tensor, seed: int, subsets_per_size: int = 8) -> dict:
    """The gate of ONE frame, for every (target, ray count): `exact` / `approx` are (H, W, 3) or (R, 3) renders of the same rays with
    the same random draws.  Returns {"self_psnr_db": ..., "cells": {target: {"whole" | "1024" | "3001": worst |dPSNR| in dB}}}; a
    subset cell is the worst of `subsets_per_size` scattered ray subsets (seeded permutations of the frame's rays)."""
    ex, ap = exact.reshape(-1, 3).double(), approx.reshape(-1, 3).double()
    n = ex.shape[0]
    tg = {k: v.reshape(-1, 3) for k, v in targets_for(ex, seed).items()}
    g = torch.Generator().manual_seed(seed + 977)
    subsets = {str(m): [torch.randperm(n, generator=g)[:m].to(ex.device) for _ in range(subsets_per_size)] for m in SUBSET_RAYS if m < n}
    cells = {}
    for name, t in tg.items():
        se_ex, se_ap = ((ex - t) ** 2).sum(dim=1), ((ap - t) ** 2).sum(dim=1)          # per-ray squared errors: subsets are gathers of these
        d = lambda a, b: abs(10.0 * math.log10(float(a) / float(b)))
        row = {"whole": d(se_ap.sum(), se_ex.sum())}
        for m, idxs in subsets.items():
            row[m] = max(d(se_ap[i].sum(), se_ex[i].sum()) for i in idxs)
        cells[name] = row
    return {"self_psnr_db": psnr(ap, ex), "cells": cells}


def worst_of(rows: list) -> dict:
    """Cell-wise worst over frames of gate_cells() results, with the lowest self-PSNR."""
    out = {"frames": len(rows), "min_self_psnr_db": min(r["self_psnr_db"] for r in rows), "max_self_psnr_db": max(r["self_psnr_db"] for r in rows),
           "cells": {}}
    for r in rows:
        for t, row in r["cells"].items():
            for m, v in row.items():
                cur = out["cells"].setdefault(t, {})
                cur[m] = max(cur.get(m, 0.0), v)
    return out


def required_self_psnr_db(target_db: float, gate_db: float = GATE_DB) -> float:
    """Self-PSNR (arithmetic vs exact frame) above which the DETERMINISTIC term of the gate, 4.34 mse(e) / mse(target), stays below
    `gate_db` for a target the exact frame approximates to `target_db` dB: target_db + 10 log10(4.343 / gate_db) = target_db + 46.4.
    (The cross term adds ~ 8.7 sqrt(mse(e) / (mse(target) 3 rays)) dB on top: it dominates on small ray sets.)"""
    return target_db + 10.0 * math.log10(10.0 / math.log(10.0) / gate_db)


# What each reduced arithmetic is EXPECTED to do at the 1e-4 dB gate, per scene, target and ray count (True: passes with >= 2x margin;
# False: misses by >= 2x; None: within 2x of the gate either way), measured on MI355X over 8 frames of bench.py's scene (x1000 density
# head) and 4 of the same scene with SURVEY 8(d)'s x40 head, against the product's exact-f32 frame: profiles/r06_gate_sensitivity.md.
# tests/test_gpu_gate.py holds the kernels to this table in BOTH directions.  Reading it:
#   * f16x3 passes every whole-frame cell on both scenes; where it misses (x1000 head, <= 3001 rays, 30 / 40 dB targets) the exact-f32
#     kernel misses by as much or more against a float64 evaluation (the scene's fp32 noise floor: self-PSNR 89-92 dB) -- no fp32
#     implementation, the reference's own on another device included, is closer to another one there;
#   * bf16x3 holds the gate against SURVEY's random target and 20 dB targets on whole frames, everything on the x40 head;
#   * f16x2 holds it on whole frames against SURVEY's random target (x1000 head) and against every target on the x40 head -- NOT against
#     a target the render approximates to 30 dB on the x1000 head (4e-3 dB: 43x the gate).
EXPECTED_PASS = {
    "f16x3": {
        "bench": {"random": {"whole": True, "1024": True, "3001": True}, "20dB": {"whole": True, "1024": None, "3001": True},
                  "30dB": {"whole": True, "1024": False, "3001": None}, "40dB": {"whole": True, "1024": False, "3001": False}},
        "soft": {"random": {"whole": True, "1024": True, "3001": True}, "20dB": {"whole": True, "1024": True, "3001": True},
                 "30dB": {"whole": True, "1024": True, "3001": True}, "40dB": {"whole": True, "1024": True, "3001": True}},
    },
    "bf16x3": {
        "bench": {"random": {"whole": True, "1024": None, "3001": None}, "20dB": {"whole": True, "1024": False, "3001": False},
                  "30dB": {"whole": None, "1024": False, "3001": False}, "40dB": {"whole": False, "1024": False, "3001": False}},
        "soft": {"random": {"whole": True, "1024": True, "3001": True}, "20dB": {"whole": True, "1024": True, "3001": True},
                 "30dB": {"whole": True, "1024": True, "3001": True}, "40dB": {"whole": True, "1024": None, "3001": None}},
    },
    "f16x2": {
        "bench": {"random": {"whole": None, "1024": False, "3001": False}, "20dB": {"whole": False, "1024": False, "3001": False},
                  "30dB": {"whole": False, "1024": False, "3001": False}, "40dB": {"whole": False, "1024": False, "3001": False}},
        "soft": {"random": {"whole": True, "1024": None, "3001": True}, "20dB": {"whole": True, "1024": None, "3001": None},
                 "30dB": {"whole": True, "1024": False, "3001": None}, "40dB": {"whole": None, "1024": False, "3001": False}},
    },
}
# arithmetics whose whole-frame gate does not hold at a 30 dB target on every scene measured: launch/eval_sharded.py verifies them on the
# user's own sequence by default (--verify-gate 50)
VERIFY_BY_DEFAULT = ("bf16x3", "f16x2")
