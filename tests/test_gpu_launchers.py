"""GPU: the two launchers end to end on a synthetic on-disk dataset (reference formats: transforms_*.json, PNGs,
bg/00050.png, index_map.npy, YAML config, checkpoint dictionary)."""
import os
import sys

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_then_eval_roundtrip(hip_lib, gpu, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))
    import make_synthetic_dataset as MS
    from launch import eval_sharded, train_sharded
    base = str(tmp_path)
    MS.write(os.path.join(base, "data"))
    cfg_path = os.path.join(base, "config.yml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(MS.config(os.path.join(base, "data"), os.path.join(base, "logs")), f)
    import contextlib, io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        logdir = train_sharded.main(["--config", cfg_path])
    log = buf.getvalue()
    print(log)
    val = [l for l in log.splitlines() if l.startswith("[VAL] Iter: 0 ")]          # TR:427-505: validation at iteration 0
    assert len(val) == 1 and 0.0 < float(val[0].split("Validation PSNR: ")[1].split()[0]) < 60.0
    # rank-0 scalars with the reference's TensorBoard tags (TR:415-424, 518-541): an event file, or scalars.jsonl without tensorboard
    import json
    sc = os.path.join(logdir, "scalars.jsonl")
    assert os.path.exists(sc) or any(f.startswith("events.out") for f in os.listdir(logdir))
    if os.path.exists(sc):
        rows = [json.loads(l) for l in open(sc)]
        tags = {r["tag"] for r in rows}
        assert {"train/coarse_loss", "train/fine_loss", "train/psnr", "train/code_loss", "validation/loss", "validation/coarse_loss",
                "validation/fine_loss", "validation/psnr"} <= tags, tags
        assert sorted(r["step"] for r in rows if r["tag"] == "train/psnr") == list(range(6))       # every iteration, flushed in batches
        assert all(np.isfinite(r["value"]) for r in rows)
    ck_path = os.path.join(logdir, "checkpoint00005.ckpt")
    assert os.path.exists(os.path.join(logdir, "checkpoint00000.ckpt")) and os.path.exists(ck_path)
    ck = torch.load(ck_path, map_location="cpu")
    assert set(ck) == {"iter", "model_coarse_state_dict", "model_fine_state_dict", "optimizer_state_dict", "loss", "psnr",
                       "background", "latent_codes"}
    assert ck["latent_codes"].shape == (6, 32) and len(ck["optimizer_state_dict"]["param_groups"]) == 2
    assert float(ck["latent_codes"].abs().sum()) > 0                     # latent rows did train
    # resume runs and keeps training the SAME latent tensor
    cfg = yaml.safe_load(open(cfg_path))
    cfg["experiment"]["train_iters"] = 8
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    train_sharded.main(["--config", cfg_path, "--load-checkpoint", ck_path])
    ck2 = torch.load(os.path.join(logdir, "checkpoint00007.ckpt"), map_location="cpu")
    assert not torch.equal(ck2["latent_codes"], ck["latent_codes"])
    # the same resume on the split-fp16 training kernels: finite loss, parameters move
    import nerf
    try:
        train_sharded.main(["--config", cfg_path, "--load-checkpoint", ck_path, "--precision", "f16x3"])
    finally:
        nerf.set_mlp_precision("f32")
    ck3 = torch.load(os.path.join(logdir, "checkpoint00007.ckpt"), map_location="cpu")
    assert np.isfinite(float(ck3["loss"])) and not torch.equal(ck3["latent_codes"], ck["latent_codes"])
    w2, w3 = ck2["model_fine_state_dict"]["layers_xyz.1.weight"], ck3["model_fine_state_dict"]["layers_xyz.1.weight"]
    assert float((w2 - w3).abs().max()) < 0.2 * float(w2.abs().max())            # a few Adam steps apart at most (Adam moves every weight by ~lr per step)
    # eval: 3 test frames -> PNGs, identical for both precisions to within quantisation
    out = os.path.join(base, "render")
    frames = eval_sharded.main(["--config", cfg_path, "--checkpoint", ck_path, "--savedir", out, "--save-disparity-image",
                                "--save-normals", "--save-error-image", "--precision", "f32"])
    err = np.asarray(__import__("PIL.Image").Image.open(os.path.join(out, "error", "0002.png")))
    assert err.shape == (32, 32, 3) and err.std() > 0                         # EV:492-497 (native resolution, jet map)
    assert np.asarray(__import__("PIL.Image").Image.open(os.path.join(out, "normals", "0000.png"))).shape == (31, 31, 3)
    assert frames == [0, 1, 2]
    from PIL import Image
    a = np.asarray(Image.open(os.path.join(out, "0001.png")))
    assert a.shape == (32, 32, 3) and a.std() > 0
    out2 = os.path.join(base, "render_b")
    eval_sharded.main(["--config", cfg_path, "--checkpoint", ck_path, "--savedir", out2, "--precision", "bf16x3"])
    b = np.asarray(Image.open(os.path.join(out2, "0001.png")))
    assert a.shape == b.shape                      # (perturb=True in validation, as shipped: images differ by sampling noise)
    # round 6: bf16x3 / f16x2 do not keep the 1e-4 dB gate against realistic targets on every scene, so the launcher verifies them by
    # default (every 50th frame of a rank, i.e. frame 0 here, also on the exact-f32 kernels) -- f16x3 and f32 are not verified
    assert eval_sharded.main.last_stats["gate"]["frames"] == 1 and eval_sharded.main.last_stats["gate"]["precision"] == "bf16x3"
    out3 = os.path.join(base, "render_h")          # split-fp16: pre-flight range probe + sticky range flag run inside the launcher
    assert eval_sharded.main(["--config", cfg_path, "--checkpoint", ck_path, "--savedir", out3, "--precision", "f16x3"]) == [0, 1, 2]
    h = np.asarray(Image.open(os.path.join(out3, "0001.png")))
    assert h.shape == a.shape and h.std() > 0 and "gate" not in eval_sharded.main.last_stats
    out4 = os.path.join(base, "render_x2")         # "f16x2" (two products per weight): same probe and flag, inference-only arithmetic
    assert eval_sharded.main(["--config", cfg_path, "--checkpoint", ck_path, "--savedir", out4, "--precision", "f16x2"]) == [0, 1, 2]
    x2 = np.asarray(Image.open(os.path.join(out4, "0001.png")))
    assert x2.shape == a.shape and x2.std() > 0 and eval_sharded.main.last_stats["gate"]["frames"] == 1
    out5 = os.path.join(base, "render_x2_verified")  # --verify-gate: every frame also on the exact-f32 kernels, same draws; the gate is REPORTED
    assert eval_sharded.main(["--config", cfg_path, "--checkpoint", ck_path, "--savedir", out5, "--precision", "f16x2", "--verify-gate", "1"]) == [0, 1, 2]
    gate = eval_sharded.main.last_stats["gate"]
    print("launcher gate check (32 x 32 frames: 1024 rays, a noisy estimate of the metric):", gate)
    assert gate["frames"] == 3 and gate["precision"] == "f16x2" and np.isfinite(gate["worst_abs_dpsnr_db"]) and gate["min_self_psnr_db"] > 40.0
    assert nerf.get_mlp_precision() == "f32"           # the launcher leaves the process-global switch as it found it
    assert os.path.exists(os.path.join(out, "disparity", "0002.png"))


def test_two_ranks_one_gpu_gloo_train_and_eval(hip_lib, gpu, tmp_path):
    """N > 1 through the launchers themselves: torch.distributed.run spawns 2 ranks of launch.train_sharded / launch.eval_sharded
    on this one GPU with the gloo backend (RCCL needs one device per rank; the launcher code, the flat gradient all-reduce and
    the kernels are the same).  Training: 3 steps, both ranks end with identical parameters (dp_consistency.json) although
    they drew different frames / rays.  Eval: the frame sets are disjoint and complete, and every PNG equals the
    single-process render byte for byte (validation sampling made deterministic for the comparison)."""
    import json
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))
    import make_synthetic_dataset as MS
    from launch import eval_sharded
    base = str(tmp_path)
    MS.write(os.path.join(base, "data"), n_test=5)
    cfgd = MS.config(os.path.join(base, "data"), os.path.join(base, "logs"), train_iters=3)
    cfgd["experiment"]["save_every"] = 2
    cfgd["nerf"]["validation"]["perturb"] = False
    cfg_path = os.path.join(base, "config.yml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfgd, f)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "4d-facial-avatars_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29600 + (os.getpid() % 1000)
    run = lambda prt: [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                       "127.0.0.1", "--master-port", str(prt), "-m"]
    r = subprocess.run(run(port) + ["launch.train_sharded", "--config", cfg_path, "--backend", "gloo"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    logdir = os.path.join(base, "logs", "synthetic")
    rep = json.load(open(os.path.join(logdir, "dp_consistency.json")))
    assert rep["world"] == 2 and rep["iters"] == 3
    assert rep["identical_parameters"], rep
    assert rep["distinct_draws"], rep                      # per-rank torch seed: the ranks do not sample the same pixels
    ck_path = os.path.join(logdir, "checkpoint00002.ckpt")
    assert os.path.exists(ck_path)
    out2 = os.path.join(base, "render2")
    r = subprocess.run(run(port + 1) + ["launch.eval_sharded", "--config", cfg_path, "--checkpoint", ck_path, "--savedir", out2,
                                        "--backend", "gloo"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=900)
    log = r.stdout.decode()
    assert r.returncode == 0, log[-3000:]
    assert "[rank 0] rendered 3 of 5 frames" in log and "[rank 1] rendered 2 of 5 frames" in log
    out1 = os.path.join(base, "render1")
    assert eval_sharded.main(["--config", cfg_path, "--checkpoint", ck_path, "--savedir", out1]) == [0, 1, 2, 3, 4]
    for i in range(5):
        a, b = open(os.path.join(out1, f"{i:04d}.png"), "rb").read(), open(os.path.join(out2, f"{i:04d}.png"), "rb").read()
        assert a == b, i
    assert sorted(os.listdir(out2)) == [f"{i:04d}.png" for i in range(5)]


def test_bench_two_ranks_one_gpu_gloo(hip_lib, gpu):
    """bench.py's N > 1 path as the driver launches it (torch.distributed.run, one JSON line from rank 0): two ranks on this one
    GPU with NERFACE_DIST_BACKEND=gloo.  Frames are sharded (weak scaling: rays_total doubles), the extras run under the same
    protocol, and the data-parallel `train` object goes through the flat gradient all-reduce."""
    import json
    import subprocess
    env = dict(os.environ, NERFACE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = 29700 + (os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--train-steps", "3"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    from tests.util import parse_bench_stdout
    c, d = parse_bench_stdout(r.stdout.decode())
    assert c["n_gpus"] == 2 and c["ranks_seen"] == 2 and c["steps"] == 1 and abs(c["value"] - d["value"]) < 1e-5 * d["value"]
    assert c["summary"]["train_bytes_allreduced"] == 4 * (2 * 552196 + 1000 * 32)
    assert d["n_gpus"] == 2 and d["steps"] == 1 and d["scaling"] == "weak" and d["dtype"] == "f32" and d["value"] > 0
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 2 * 512 * 512) < 1.0                    # two ranks x one frame each
    assert "split_f16" in d and "split_bf16" in d and "cpu_baseline" not in d
    assert set(k for k in d["train"] if k not in ("workload", "allreduce")) == {"f32", "f16x3", "bf16x3"}
    # evidence of the N > 1 path on the line itself: ranks the communicator saw, bytes and time of the flat all-reduce
    ar = d["train"]["allreduce"]
    assert d["ranks_seen"] == 2 and ar["ranks_seen"] == 2 and ar["backend"] == "gloo" and ar["allreduce_us"] > 0
    assert ar["bytes_allreduced"] == 4 * (2 * 552196 + 1000 * 32)          # live parameters of both models + the latent table
    assert d["summary"]["train_bytes_allreduced"] == ar["bytes_allreduced"] and d["summary"]["ranks_seen"] == 2
    assert "data parallel over 2 GPUs" in d["train"]["workload"]
    assert d["roofline"]["traffic"] is None                                                     # PMC passes are an N = 1 extra


def test_bench_bare_gpus_2_launches_its_own_ranks(hip_lib, gpu):
    """A bare `python bench.py --gpus 2` (no launcher, WORLD_SIZE unset) starts the two ranks itself and reports them: n_gpus == 2,
    ranks_seen == 2 on the compact line (it used to measure one GPU and say so nowhere).  Two ranks on this one
    GPU with NERFACE_DIST_BACKEND=gloo.  A launcher whose world size disagrees with --gpus is refused."""
    import subprocess
    from tests.util import parse_bench_stdout
    env = dict(os.environ, NERFACE_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extras"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    c, d = parse_bench_stdout(r.stdout.decode())
    assert c["n_gpus"] == 2 and c["ranks_seen"] == 2 and c["summary"]["ranks_seen"] == 2
    assert abs(c["value"] * c["ms_per_step"] * 1e-3 - 2 * 512 * 512) < 30.0                    # two ranks x one frame each (6 significant digits)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras"],
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"must agree" in r.stderr
    # per-rank step times on the line: stragglers show
    assert len(d["per_rank_ms_per_step"]) == 2 and c["summary"]["rank_ms_max"] >= c["summary"]["rank_ms_min"] > 0
    assert c["summary"]["rank_ms_max"] <= c["ms_per_step"] * (1 + 1e-6)           # a rank's own frames, measured before the closing barrier


def test_bench_more_ranks_than_devices_fails_fast_with_the_reason(hip_lib, gpu):
    """`bench.py --gpus N` with the nccl (RCCL) backend on a box with fewer than N devices must exit non-zero within a
    minute with a message naming device_count -- not hang in the rendezvous or inside ncclCommInitRank.  Both entry forms: the bare
    command (it would launch its own ranks) and a rank started by a launcher."""
    import subprocess
    import time
    import torch
    n = torch.cuda.device_count() + 1
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "NERFACE_DIST_BACKEND"):
        env.pop(k, None)
    for extra in ({}, {"WORLD_SIZE": str(n), "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"}):
        t0 = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-extras"],
                           env=dict(env, **extra), cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert r.returncode != 0 and b"device_count" in r.stderr, r.stderr.decode()[-2000:]
        assert time.time() - t0 < 60.0


def _torchrun(n, port):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port)]


def test_rccl_single_rank_runs_every_collective(hip_lib, gpu, tmp_path):
    """RCCL before an 8-GPU node ever runs it (VERDICT r02 #3a): one rank under torch.distributed.run with the **nccl** backend and
    NERFACE_DIST_FORCE=1, which brings the process group up at world size 1 (RCCL load, `device_id=` init, dmabuf IPC setting)
    and makes the trainer run the collectives it would skip: the parameter broadcast, the has-gradient negotiation, one flat
    all-reduce of the gradients on device memory per step, the validation all-reduce, the final all-gather.  Through the
    launcher and through bench.py --mode train as the driver launches it."""
    import json
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_synthetic_dataset as MS
    base = str(tmp_path)
    MS.write(os.path.join(base, "data"))
    cfgd = MS.config(os.path.join(base, "data"), os.path.join(base, "logs"), train_iters=3)
    cfg_path = os.path.join(base, "config.yml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfgd, f)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "4d-facial-avatars_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""),
               HSA_ENABLE_IPC_MODE_LEGACY="0", NERFACE_DIST_FORCE="1")
    env.pop("NERFACE_DIST_BACKEND", None)
    port = 29800 + (os.getpid() % 1000)
    r = subprocess.run(_torchrun(1, port) + ["-m", "launch.train_sharded", "--config", cfg_path, "--backend", "nccl"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    log = r.stdout.decode()
    assert r.returncode == 0, log[-3000:]
    rep = json.load(open(os.path.join(base, "logs", "synthetic", "dp_consistency.json")))
    assert rep["world"] == 1 and rep["iters"] == 3 and rep["identical_parameters"]
    assert "[DP] 1 ranks" in log and "[VAL] Iter: 0" in log
    r = subprocess.run(_torchrun(1, port + 1) + [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--mode", "train", "--steps", "3", "--warmup", "1"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    from tests.util import parse_bench_stdout
    c, d = parse_bench_stdout(r.stdout.decode())
    assert c["n_gpus"] == 1 and c["ranks_seen"] == 1 and c["roofline"]["bound"] == "mfma"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["roofline"]["bound"] == "mfma" and len(d["roofline"]["kernels"]) == 3
    # RCCL really ran the flat all-reduce on device memory: the line says how many ranks the communicator saw, what moved, how long it took
    assert d["ranks_seen"] == 1 and d["allreduce"]["backend"] == "nccl" and d["allreduce"]["calls"] == 4
    assert d["bytes_allreduced"] == 4 * (2 * 552196 + 1000 * 32) and d["allreduce_us"] > 0
    assert d["summary"]["allreduce_us"] == d["allreduce_us"]


def test_eight_ranks_one_gpu_gloo(hip_lib, gpu, tmp_path):
    """Rank-count-dependent paths at the width of the target node (VERDICT r02 #3b): 8 ranks on this one GPU (gloo).  Training: 2
    validation frames over 8 ranks (6 ranks contribute zero to the all-reduced validation loss), 8 distinct ray draws, identical
    parameters on all ranks after 2 steps.  Eval: 5 test frames over 8 ranks -- ranks 5..7 render nothing and still reach the
    barrier; the frame set is complete.  bench.py: --gpus 8 frame shard, and the data-parallel training step at 8."""
    import json
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_synthetic_dataset as MS
    base = str(tmp_path)
    MS.write(os.path.join(base, "data"), n_test=5)
    cfgd = MS.config(os.path.join(base, "data"), os.path.join(base, "logs"), train_iters=2)
    cfgd["experiment"]["save_every"] = 1
    cfg_path = os.path.join(base, "config.yml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfgd, f)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "4d-facial-avatars_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""),
               HSA_ENABLE_IPC_MODE_LEGACY="0", NERFACE_DIST_BACKEND="gloo")
    env.pop("NERFACE_DIST_FORCE", None)
    port = 29900 + (os.getpid() % 1000)
    r = subprocess.run(_torchrun(8, port) + ["-m", "launch.train_sharded", "--config", cfg_path, "--backend", "gloo"], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    logdir = os.path.join(base, "logs", "synthetic")
    rep = json.load(open(os.path.join(logdir, "dp_consistency.json")))
    assert rep["world"] == 8 and rep["identical_parameters"] and rep["distinct_draws"], rep
    ck_path = os.path.join(logdir, "checkpoint00001.ckpt")
    out = os.path.join(base, "render8")
    r = subprocess.run(_torchrun(8, port + 1) + ["-m", "launch.eval_sharded", "--config", cfg_path, "--checkpoint", ck_path, "--savedir", out,
                                                  "--backend", "gloo"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       timeout=1200)
    log = r.stdout.decode()
    assert r.returncode == 0, log[-3000:]
    assert sorted(os.listdir(out)) == [f"{i:04d}.png" for i in range(5)]
    for k in range(5):
        assert f"[rank {k}] rendered 1 of 5 frames" in log
    r = subprocess.run(_torchrun(8, port + 2) + [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0", "--no-extras"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    from tests.util import parse_bench_stdout
    c, d = parse_bench_stdout(r.stdout.decode())
    assert c["n_gpus"] == 8 and c["ranks_seen"] == 8
    assert d["n_gpus"] == 8 and abs(d["value"] * d["ms_per_step"] * 1e-3 - 8 * 512 * 512) < 1.0
    r = subprocess.run(_torchrun(8, port + 3) + [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--mode", "train", "--steps", "2", "--warmup", "1"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    c, d = parse_bench_stdout(r.stdout.decode())
    assert c["n_gpus"] == 8 and c["ranks_seen"] == 8 and c["summary"]["ranks_seen"] == 8
    assert d["n_gpus"] == 8 and d["config"]["rays_per_step"] == 8 * 2048 and d["value"] > 0
    assert d["ranks_seen"] == 8 and d["allreduce"]["calls"] == 3 and d["bytes_allreduced"] == 4 * (2 * 552196 + 1000 * 32) and d["allreduce_us"] > 0


def test_launchers_second_model_family(hip_lib, gpu, tmp_path):
    """The same two launchers with `type: ConditionalBlendshapeLearnableCodeNeRFModel` in the config (as 6 shipped configs
    have): trains (exact-f32 kernels), checkpoints with this family's state_dict keys, renders in both precisions."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "4d-facial-avatars_amd"))
    import make_synthetic_dataset as MS
    from launch import eval_sharded, train_sharded
    from oracle import nerface_oracle as O
    from PIL import Image
    base = str(tmp_path)
    MS.write(os.path.join(base, "data"))
    cfg_path = os.path.join(base, "config.yml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(MS.config(os.path.join(base, "data"), os.path.join(base, "logs"),
                                 model_type="ConditionalBlendshapeLearnableCodeNeRFModel"), f)
    logdir = train_sharded.main(["--config", cfg_path])
    ck0 = torch.load(os.path.join(logdir, "checkpoint00000.ckpt"), map_location="cpu")
    ck = torch.load(os.path.join(logdir, "checkpoint00005.ckpt"), map_location="cpu")
    assert list(ck["model_fine_state_dict"].keys()) == O.LCODE_KEYS
    moved = [k for k in O.LCODE_KEYS if not torch.equal(ck["model_fine_state_dict"][k], ck0["model_fine_state_dict"][k])]
    assert len(moved) == len(O.LCODE_KEYS), set(O.LCODE_KEYS) - set(moved)        # every tensor of the family receives gradients
    assert float(ck["latent_codes"].abs().sum()) > 0 and np.isfinite(float(ck["loss"]))
    for prec in ("f32", "bf16x3", "f16x3", "f16x2"):
        out = os.path.join(base, "render_" + prec)
        assert eval_sharded.main(["--config", cfg_path, "--checkpoint", os.path.join(logdir, "checkpoint00005.ckpt"), "--savedir", out,
                                  "--precision", prec]) == [0, 1, 2]
        a = np.asarray(Image.open(os.path.join(out, "0001.png")))
        assert a.shape == (32, 32, 3) and a.std() > 0


def test_eval_postprocess_against_reference_fixture(hip_lib, gpu):
    """f3 on the device: k_eval_postprocess against the outputs of the UNMODIFIED eval script (cast_to_image EV:184-190,
    torch_normal_map(clean=True) EV:84-119; tests/golden/eval_post.npz).  Both outputs are byte-exact: the kernel executes the
    same IEEE operation sequence (no contraction) as the reference's tensor ops."""
    from nerf import ops
    from oracle import make_golden as MG
    from oracle import nerface_oracle as O
    g = np.load(os.path.join(ROOT, "tests", "golden", "eval_post.npz"))
    for n, (rgb, disp, w) in MG.eval_post_inputs().items():
        u8, nrm = ops.eval_postprocess(rgb.to(gpu), disp.to(gpu), w.to(gpu), O.INTRINSICS)
        assert np.array_equal(u8.cpu().numpy(), g[f"rgb_u8_{n}"])
        d = np.abs(nrm.cpu().numpy().astype(int) - g[f"normals_u8_{n}"].astype(int))
        print(f"normal map {n}x{n}: {int((d != 0).sum())} of {d.size} bytes differ (max {int(d.max())})")
        assert nrm.shape == (n - 1, n - 1, 3) and int(d.max()) == 0
        _, plain = ops.eval_postprocess(rgb.to(gpu), disp.to(gpu), None, O.INTRINSICS)
        d = np.abs(plain.cpu().numpy().astype(int) - g[f"normals_plain_u8_{n}"].astype(int))
        assert int(d.max()) == 0
