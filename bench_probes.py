"""Probe legs of bench.py, all outside every timed region: what the box is (`device_info`, `power_probe`, `pattern_store_probe`), the
rocprofv3 PMC passes behind `roofline.traffic` / the sustained clocks (`pmc_*`), and the launcher throughput leg (`launcher_eval_leg`)."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "4d-facial-avatars_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bench_common import *  # noqa: E402,F401,F403


def launcher_eval_leg(dev, model_c, model_f, n_frames=32):
    """configs[3] readiness: launch/eval_sharded.py itself on a synthetic 512x512 sequence of n_frames test frames in the on-disk
    format (tools/make_synthetic_dataset.py), one GPU, f32, PNG + normal-map output -- frames/s of the loop's WALL time (including
    the PNG tail) against the GPU seconds per frame its HIP events measure: wall / GPU ~ 1 means the sequence render is not
    host-bound (EV:392-498 with EV:42-51, 469-488 moved to the device / to worker threads)."""
    import shutil
    import tempfile
    import yaml
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_synthetic_dataset as MS
    from launch import eval_sharded
    base = tempfile.mkdtemp(prefix="nf_launcher_")
    try:
        data = MS.write(os.path.join(base, "data"), size=H, n_train=2, n_val=1, n_test=n_frames)
        cfg = MS.config(data, os.path.join(base, "logs"))
        cfg["nerf"]["validation"].update(num_coarse=N_COARSE, num_fine=N_FINE, chunksize=CHUNK)
        cfg_path = os.path.join(base, "config.yml")
        with open(cfg_path, "w") as f:
            yaml.safe_dump(cfg, f)
        ck_path = os.path.join(base, "ck.ckpt")
        torch.save({"model_coarse_state_dict": model_c.state_dict(), "model_fine_state_dict": model_f.state_dict(),
                    "latent_codes": 0.1 * torch.randn(2, 32), "background": None}, ck_path)
        out = os.path.join(base, "render")
        eval_sharded.main(["--config", cfg_path, "--checkpoint", ck_path, "--savedir", out, "--save-normals", "--precision", "f32"])
        st = dict(eval_sharded.main.last_stats)
        n_png = len([f for f in os.listdir(out) if f.endswith(".png")])
        assert n_png == n_frames, (n_png, n_frames)
        return {"launcher_eval_frames_s": st["frames_s"], "launcher_gpu_s_per_frame": st["gpu_s_per_frame"],
                "launcher_wall_over_gpu": st["wall_s"] / st["gpu_s_total"], "frames": st["frames"], "wall_s": st["wall_s"],
                "wall_s_until_gpu_idle": st["wall_s_until_gpu_idle"],
                "what": f"launch/eval_sharded.py, {n_frames} test frames 512x512, 64+128, f32, PNG + normals written, one GPU; wall includes the PNG tail"}
    finally:
        shutil.rmtree(base, ignore_errors=True)


def pattern_store_probe():
    """tools/micro/store_bw: the training kernels' store pattern (256 persistent workgroups, 1 KiB per instruction, nine 256 MiB
    planes = 2.4 GB, nt and default policy) with nothing else in the way -- separates a box whose memory system takes these
    streams badly from a good one, which the sequential fill probe does not."""
    exe = os.path.join(ROOT, "tools", "micro", "store_bw")
    if not os.path.exists(exe):
        return {"error": "tools/micro/store_bw not built (__graft_entry__.build())"}
    try:
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        return {"error": repr(e)}


def device_info(dev):
    """What the box reports (re-derive the peaks from the clocks of the GPU box): CUs x 256 fp32-MFMA FLOP per clock x
    the engine clock, beside the vendor figure the roofline is priced against."""
    p = torch.cuda.get_device_properties(dev)
    mhz = float(getattr(p, "clock_rate", 0)) / 1e3
    if mhz <= 0:                                                        # torch on ROCm reports no clock: ask rocminfo (gfx agent's max clock)
        try:
            txt = subprocess.run(["rocminfo"], capture_output=True, text=True, timeout=20).stdout
            blocks = [b for b in txt.split("*******") if "gfx950" in b and "Max Clock Freq" in b]
            if blocks:
                mhz = float(re.search(r"Max Clock Freq\. \(MHz\):\s*(\d+)", blocks[0]).group(1))
        except Exception:
            mhz = 0.0
    cus = int(p.multi_processor_count)
    # what THIS box's HBM does right now (GPU boxes of the pool differ: one ran every store-heavy kernel 2x slower, profiles/r03_experiments.md §7):
    # a 1 GiB fill (pure writes) and a 1 GiB copy (read + write) with torch's own kernels, best of 5
    probe = {}
    try:
        x = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        y = torch.empty_like(x)
        for name, fn, nbytes in (("hbm_fill_gbs", lambda: x.fill_(1.0), x.numel() * 4), ("hbm_copy_gbs", lambda: y.copy_(x), 2 * x.numel() * 4)):
            best = 0.0
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                best = max(best, nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
            probe[name] = best
        del x, y
        torch.cuda.empty_cache()
    except Exception as e:
        probe = {"hbm_probe_error": repr(e)}
    probe["env"] = {k: v for k, v in os.environ.items() if re.match(r"(HSA|HIP|ROCR|ROCM|GPU|AMD|PYTORCH|NCCL|RCCL)_", k)}   # what differs box to box
    return {**probe, "name": p.name, "arch": getattr(p, "gcnArchName", ""), "compute_units": cus, "engine_clock_mhz": mhz,
            "hbm_gib": round(p.total_memory / 2 ** 30, 1),
            "fp32_mfma_peak_from_clock_tflops": cus * 256 * mhz * 1e6 / 1e12, "fp32_mfma_peak_priced_tflops": PEAK_F32_MFMA_TFLOPS}


def _gpu_sysfs(dev):
    """sysfs directory of the amdgpu device `dev` runs on (None if the container does not show it)."""
    import glob
    cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(d, "power_dpm_force_performance_level")))
    if not cards:
        return None
    try:                                                             # match by PCI bus id when torch reports it
        want = torch.cuda.get_device_properties(dev).pci_bus_id
        for d in cards:
            if int(os.path.basename(os.path.realpath(d)).split(":")[1], 16) == want:
                return d
    except Exception:
        pass
    return cards[min(dev.index or 0, len(cards) - 1)]


def power_probe(dev, legs, seconds=1.5, period=0.02):
    """What the board's power management does to each inference kernel on THIS box: socket power, engine / fabric / memory clock read
    from amdgpu's sysfs nodes every 20 ms while the kernel runs back to back for `seconds`, beside the board's power cap and performance
    level.  Boxes of the pool differ in how they hold the cap: the builder's lower the engine clock under the 16-bit MFMA kernels
    (2.1-2.2 GHz), the driver's boxes of rounds 3 and 4 reported 2.38 GHz for every kernel and 1.35x (inference) to 2.9x (training
    forward) the busy cycles.  Outside every timed region; reads only."""
    import glob, threading
    d = _gpu_sysfs(dev)
    if d is None:
        return {"error": "no amdgpu sysfs node visible"}
    hw = (glob.glob(os.path.join(d, "hwmon", "hwmon*")) or [None])[0]

    def rd(path, num=True):
        try:
            t = open(path).read().strip()
            return float(t) if num else t
        except Exception:
            return None
    pwr = next((f for f in ("power1_average", "power1_input") if hw and os.path.exists(os.path.join(hw, f))), None)
    static = {"sysfs": d, "perf_level": rd(os.path.join(d, "power_dpm_force_performance_level"), False),
              "power_cap_w": (rd(os.path.join(hw, "power1_cap")) or 0) / 1e6 if hw else None,
              "power_cap_max_w": (rd(os.path.join(hw, "power1_cap_max")) or 0) / 1e6 if hw else None,
              "power_node": pwr}
    for node in ("pp_dpm_sclk", "pp_dpm_fclk", "pp_dpm_mclk", "current_compute_partition", "current_memory_partition"):
        static[node] = rd(os.path.join(d, node), False)

    def star(node):                                                  # the level amdgpu marks as current in a pp_dpm_* table (MHz)
        t = rd(os.path.join(d, node), False) or ""
        m = re.search(r"(\d+)\s*Mhz\s*\*", t, re.I)
        return float(m.group(1)) if m else None
    out = {"static": static}
    for name, fn in legs:
        fn()
        torch.cuda.synchronize()
        rows, stop = [], threading.Event()

        def sample():
            while not stop.is_set():
                rows.append((rd(os.path.join(hw, pwr)) if pwr else None, rd(os.path.join(hw, "freq1_input")) if hw else None,
                             star("pp_dpm_sclk"), star("pp_dpm_fclk"), star("pp_dpm_mclk")))
                time.sleep(period)
        th = threading.Thread(target=sample, daemon=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n, t0 = 0, time.perf_counter()
        th.start()
        e0.record()
        while time.perf_counter() - t0 < seconds:
            fn()
            n += 1
            if n % 4 == 0:
                torch.cuda.synchronize()                             # keep the queue short: the loop ends on time
        e1.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        rows = rows[len(rows) // 4:]                                  # the first quarter is the ramp
        mean = lambda k, sc: (sum(r[k] for r in rows if r[k] is not None) / max(1, sum(r[k] is not None for r in rows)) * sc
                              if any(r[k] is not None for r in rows) else None)
        out[name] = {"launch_ms": e0.elapsed_time(e1) / n, "launches": n, "samples": len(rows), "power_w": mean(0, 1e-6),
                     "power_w_max": max((r[0] for r in rows if r[0] is not None), default=0) * 1e-6 if pwr else None,
                     "sclk_mhz_hwmon": mean(1, 1e-6), "sclk_mhz_dpm": mean(2, 1.0), "fclk_mhz_dpm": mean(3, 1.0), "mclk_mhz_dpm": mean(4, 1.0)}
    return out


def _pmc_guard():
    """rocprofv3 path, or (None, reason).  Never nest profilers: a PMC pass started from a process that is itself being traced
    combines counter collection with tracing -- the combination this pool's nodes do not survive."""
    import shutil
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, {"error": "rocprofv3 not found"}
    under = [k for k in os.environ if k.startswith(("ROCPROF", "ROCP_", "ROCTRACER", "ROCPROFILER"))]
    if under or "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return None, {"skipped": "bench.py is running under a profiler (" + ", ".join(sorted(under)[:4]) + "); PMC passes not nested"}
    return prof, None


def pmc_pass_rows(prof, tmp, counter, script, argv, timeout):
    """One `rocprofv3 --kernel-trace --pmc <counter>` pass over tools/<script> <argv>: [(kernel_name, grid_x, value, duration_ns
    or None)] per dispatch, or (None, detail).  The dispatch duration comes from the same pass (the counters view's own
    start / end stamps when it has them, else the kernel trace joined on the dispatch id)."""
    import sqlite3
    out = os.path.join(tmp, counter)
    env = dict(os.environ, TMPDIR=tmp)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    cmd = [prof, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "p", "--", sys.executable, os.path.join(ROOT, "tools", script), *argv]
    r = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    dbs = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
    if r.returncode != 0 or not dbs:
        return None, {"error": f"rocprofv3 pass {counter} failed (rc {r.returncode})", "tail": r.stdout.decode()[-400:]}
    con = sqlite3.connect(dbs[0])
    cols = [c[1] for c in con.execute("pragma table_info(counters_collection)").fetchall()]
    if "start" in cols and "end" in cols:
        rows = con.execute('select kernel_name, grid_size_x, value, "end" - "start" from counters_collection where counter_name = ?',
                           (counter,)).fetchall()
    elif "dispatch_id" in cols:
        try:
            kcols = [c[1] for c in con.execute("pragma table_info(kernels)").fetchall()]
            key = "dispatch_id" if "dispatch_id" in kcols else "id"
            rows = con.execute(f"select c.kernel_name, c.grid_size_x, c.value, k.duration from counters_collection c left join kernels k "
                               f"on k.{key} = c.dispatch_id where c.counter_name = ?", (counter,)).fetchall()
        except Exception:
            rows = [(n, g, v, None) for n, g, v in con.execute(
                "select kernel_name, grid_size_x, value from counters_collection where counter_name = ?", (counter,)).fetchall()]
    else:
        rows = [(n, g, v, None) for n, g, v in con.execute(
            "select kernel_name, grid_size_x, value from counters_collection where counter_name = ?", (counter,)).fetchall()]
    return rows, None


def pmc_kernel_bytes(script, argv, kernels, timeout=240):
    """HBM bytes per launch of the named kernels, measured by THIS command: two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    cannot share a pass on gfx950, MI355X_MICROARCH.md) over tools/<script> <argv>.  Per kernel (substring match, its largest grid):
    {"fetch_bytes", "write_bytes"}, the counters' KiB x 1024, raw.  Returns (dict or None, detail)."""
    import shutil
    import tempfile
    prof, why = _pmc_guard()
    if prof is None:
        return None, why
    got = {k: {} for k in kernels}
    tmp = tempfile.mkdtemp(prefix="nf_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            rows, err = pmc_pass_rows(prof, tmp, counter, script, argv, timeout)
            if rows is None:
                return None, err
            for kernel in kernels:
                hits = [(gx, v) for n, gx, v, _ in rows if kernel in n]
                big = max((gx for gx, _ in hits), default=None)           # the launch of interest is the kernel's largest grid
                vals = [v for gx, v in hits if gx == big]
                if not vals:
                    return None, {"error": f"kernel {kernel} not found in the {counter} pass", "kernels": sorted({n[:60] for n, _, _, _ in rows})[:8]}
                got[kernel]["fetch_bytes" if counter == "FETCH_SIZE" else "write_bytes"] = sum(vals) / len(vals) * 1024.0
        return got, {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) run by bench.py on tools/{script} "
                               + " ".join(argv) + " in this run"}
    except Exception as e:
        return None, {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_sustained_clock(precision="f32", timeout=240):
    """The engine clock the headline kernel actually held: GRBM_GUI_ACTIVE (busy cycles of the graphics engine) of its fine-pass launch
    divided by the duration of the same dispatch, from one rocprofv3 PMC pass over tools/pmc_one_launch.py.  rocprofv3 sums the
    counter over the 8 XCDs of the device (profiles/r01_mlp_kernels_pmc.md: 18.7e9 'cycles' per second), so a quotient above 6 GHz
    is divided by the XCD count.  Returns (MHz or None, detail)."""
    import shutil
    import tempfile
    prof, why = _pmc_guard()
    if prof is None:
        return None, why
    kernel = {"bf16x3": "k_paper_mlp_fwd_bf16", "f16x3": "k_paper_mlp_fwd_f16", "f16x2": "k_paper_mlp_fwd_f16x2"}.get(precision, "k_paper_mlp_fwd<")
    tmp = tempfile.mkdtemp(prefix="nf_pmc_")
    try:
        rows, err = pmc_pass_rows(prof, tmp, "GRBM_GUI_ACTIVE", "pmc_one_launch.py", [precision], timeout)
        if rows is None:
            return None, err
        hits = [(v, d) for n, _, v, d in rows if kernel in n and d]
        if not hits:
            return None, {"error": "no dispatch of the kernel with a duration in the GRBM_GUI_ACTIVE pass",
                          "kernels": sorted({n[:60] for n, _, _, _ in rows})[:8]}
        per = sorted(v / (d * 1e-9) for v, d in hits)
        hz = per[len(per) // 2]
        div = 8 if hz > 6e9 else 1
        return hz / div / 1e6, {"source": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE run by bench.py on tools/pmc_one_launch.py " + precision,
                                "kernel": kernel, "dispatches": len(hits), "busy_cycles_raw_median": sorted(v for v, _ in hits)[len(hits) // 2],
                                "dispatch_ms_under_pmc_median": sorted(d for _, d in hits)[len(hits) // 2] / 1e6, "xcd_sum_divisor": div}
    except Exception as e:
        return None, {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic(precision, timeout=240):
    """HBM bytes per fine-pass MLP launch of the eval kernel of `precision` (tools/pmc_one_launch.py launches exactly the kernel
    the roofline object times).  Raw counters, no 2x correction (the dominant reads are 4-byte z loads, not the 16 B/lane stream
    the guide's correction is calibrated on).  Returns (bytes or None, detail dict)."""
    kernel = {"bf16x3": "k_paper_mlp_fwd_bf16", "f16x3": "k_paper_mlp_fwd_f16", "f16x2": "k_paper_mlp_fwd_f16x2"}.get(precision, "k_paper_mlp_fwd<")
    got, detail = pmc_kernel_bytes("pmc_one_launch.py", [precision], [kernel], timeout)
    if got is None:
        return None, detail
    detail.update(got[kernel])
    return got[kernel]["fetch_bytes"] + got[kernel]["write_bytes"], detail


def pmc_train_traffic(train, timeout=300):
    """Fill `traffic` of every training kernel of the `train` object (all arithmetics in one pair of PMC passes): raw FETCH_SIZE +
    WRITE_SIZE bytes per launch."""
    precs = [p for p in ("f32", "f16x3", "bf16x3") if p in train and isinstance(train[p].get("roofline"), dict) and "kernels" in train[p]["roofline"]]
    names = [k for p in precs for k, _ in TRAIN_KERNELS[p]]
    got, detail = pmc_kernel_bytes("pmc_train_launch.py", precs, names, timeout)
    for p in precs:
        for (kname, _), obj in zip(TRAIN_KERNELS[p], train[p]["roofline"]["kernels"]):
            if got is None:
                obj["traffic_detail"] = detail
                continue
            f, w = got[kname]["fetch_bytes"], got[kname]["write_bytes"]
            # `traffic` = the RAW counters (FETCH_SIZE + WRITE_SIZE, KiB x 1024): an independent measurement.  The guide's gfx950 note
            # (FETCH_SIZE under-counts 16 B/lane reads by 2x) applies to part of these kernels' reads only (the LDS-DMA'd fragment
            # streams; dZ / d_raw rows are 4 B/lane loads), so the x2 figure is kept beside it as an ESTIMATE of the upper bound, not
            # as the measurement (calibrating the counter against the expected bytes is no measurement)
            obj["traffic"] = f + w
            obj["traffic_detail"] = {"fetch_bytes_raw": f, "write_bytes": w, "traffic_if_all_fetches_undercount_2x_estimate": 2 * f + w,
                                     "algorithmic_bytes_per_launch": obj["algorithmic_hbm_bytes_per_point"] * 2048 * 128, **detail}


def pmc_train_clocks(train, timeout=300):
    """Engine clock each training kernel actually held (GRBM_GUI_ACTIVE / dispatch time, one PMC pass over tools/pmc_train_launch.py in all
    arithmetics) and its busy cycles: the split kernels (dense 16-bit MFMA + 9 KB/point of HBM traffic) are clocked down by the power
    management to 1.55-2.1 GHz under sustained load (profiles/r04_experiments.md), so their wall time is cycles / granted clock --
    `ms_at_nominal_clock` is what the same cycles take at the 2.4 GHz the peaks are quoted at."""
    import shutil
    import tempfile
    precs = [p for p in ("f32", "f16x3", "bf16x3") if p in train and isinstance(train[p].get("roofline"), dict) and "kernels" in train[p]["roofline"]]
    prof, why = _pmc_guard()
    if prof is None or not precs:
        return why
    tmp = tempfile.mkdtemp(prefix="nf_pmc_")
    try:
        rows, err = pmc_pass_rows(prof, tmp, "GRBM_GUI_ACTIVE", "pmc_train_launch.py", precs, timeout)
        if rows is None:
            return err
        for p in precs:
            for (kname, _), obj in zip(TRAIN_KERNELS[p], train[p]["roofline"]["kernels"]):
                hits = [(v, d) for n, _, v, d in rows if kname in n and d]
                if not hits:
                    continue
                div = 8 if sorted(v / (d * 1e-9) for v, d in hits)[len(hits) // 2] > 6e9 else 1
                clk = sorted(v / div / (d * 1e-9) / 1e6 for v, d in hits)
                cyc = sorted(v / div for v, _ in hits)[len(hits) // 2]
                obj["sustained_clock_mhz"] = clk[len(clk) // 2]
                obj["sustained_clock_mhz_range"] = [clk[0], clk[-1]]
                obj["busy_mcycles"] = cyc / 1e6
                obj["ms_at_nominal_clock"] = cyc / 2.4e9 * 1e3
        return {"source": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE run by bench.py on tools/pmc_train_launch.py " + " ".join(precs)}
    except Exception as e:
        return {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
