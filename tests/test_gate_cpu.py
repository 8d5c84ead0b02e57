"""CPU: nerf/gate.py -- the arithmetic of north_star's gate (targets the frame approximates to a chosen PSNR, cells per target and ray
count, the self-PSNR an arithmetic needs) on synthetic frames; bench.py still exports what tools/ and tests/ import from it after the
round-6 split into bench_common / bench_baselines / bench_probes.  No GPU, no library."""
import importlib.util
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gate():
    spec = importlib.util.spec_from_file_location("nf_gate_under_test", os.path.join(ROOT, "4d-facial-avatars_amd", "nerf", "gate.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_targets_are_calibrated_and_seeded():
    G = _gate()
    frame = torch.rand((96, 96, 3), generator=torch.Generator().manual_seed(1))
    for db in G.TARGET_DBS:
        t = G.target_near(frame, db, seed=5)
        assert float(t.min()) >= 0.0 and float(t.max()) <= 1.0
        assert abs(G.psnr(frame, t) - db) < 5e-3, (db, G.psnr(frame, t))
        assert torch.equal(t, G.target_near(frame, db, seed=5))                  # same seed, same target (CPU generator)
    tg = G.targets_for(frame, seed=9)
    assert list(tg) == ["random", "20dB", "30dB", "40dB"] and 7.0 < G.psnr(frame, tg["random"]) < 9.5


def test_gate_cells_follow_the_closed_form():
    """dPSNR ~ 4.34 (mse(e) + 2 mean(e r)) / mse(r): with a white error of 70 dB self-PSNR the whole-frame cell at a 30 dB target is
    dominated by the deterministic term 4.34 mse(e) / mse(target) = 4.3e-4 dB, the random target is ~170x less sensitive, small ray sets
    are worse than the whole frame, and an exact copy scores zero everywhere."""
    G = _gate()
    frame = torch.rand((128, 128, 3), generator=torch.Generator().manual_seed(2)).double()
    e = 10.0 ** (-70.0 / 20.0) * torch.randn(frame.shape, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    r = G.gate_cells(frame, frame + e, seed=4)
    assert abs(r["self_psnr_db"] - 70.0) < 0.05
    det = 10.0 / math.log(10.0) * 1e-7 / 1e-3                                    # 4.34 * mse(e) / mse(target at 30 dB)
    assert 0.5 * det < r["cells"]["30dB"]["whole"] < 2.0 * det, (r["cells"]["30dB"]["whole"], det)
    assert r["cells"]["random"]["whole"] < r["cells"]["30dB"]["whole"] / 20.0
    assert r["cells"]["40dB"]["whole"] > 5.0 * r["cells"]["30dB"]["whole"]
    assert r["cells"]["30dB"]["1024"] > r["cells"]["30dB"]["whole"]
    z = G.gate_cells(frame, frame.clone(), seed=4)
    assert all(v == 0.0 for row in z["cells"].values() for v in row.values()) and z["self_psnr_db"] == float("inf")
    w = G.worst_of([r, z])
    assert w["frames"] == 2 and w["cells"]["30dB"]["whole"] == r["cells"]["30dB"]["whole"] and w["min_self_psnr_db"] == r["self_psnr_db"]
    assert abs(G.required_self_psnr_db(30.0) - 76.38) < 0.01


def test_expectation_table_is_complete():
    G = _gate()
    for prec in ("f16x3", "bf16x3", "f16x2"):
        for scene in ("bench", "soft"):
            for t in ("random", "20dB", "30dB", "40dB"):
                assert set(G.EXPECTED_PASS[prec][scene][t]) == {"whole", "1024", "3001"}
                assert all(v in (True, False, None) for v in G.EXPECTED_PASS[prec][scene][t].values())
    # the claims the docs make, as the table states them: f16x3 passes every whole-frame cell; f16x2 misses 30 dB on the x1000 head
    assert all(G.EXPECTED_PASS["f16x3"][s][t]["whole"] is True for s in ("bench", "soft") for t in ("random", "20dB", "30dB", "40dB"))
    assert G.EXPECTED_PASS["f16x2"]["bench"]["30dB"]["whole"] is False and G.EXPECTED_PASS["f16x2"]["soft"]["30dB"]["whole"] is True
    assert set(G.VERIFY_BY_DEFAULT) == {"bf16x3", "f16x2"}


def test_bench_still_exports_what_tools_import():
    sys.path.insert(0, ROOT)
    import bench
    for name in ("synth_params", "frame_pose", "options", "INTRINSICS", "NEAR", "FAR", "CHUNK", "N_COARSE", "N_FINE", "train_roofline", "bench_train",
                 "power_probe", "pmc_pass_rows", "pmc_sustained_clock", "device_info", "_pmc_guard", "cpu_baseline", "eager_rocm_reference",
                 "summary_of", "compact_line", "COMPACT_LIMIT", "SPLIT_MFMAS_PER_TILE"):
        assert hasattr(bench, name), name
    assert bench.SPLIT_MFMAS_PER_TILE == {"x3": 2982, "x2": 1988}
