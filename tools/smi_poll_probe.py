"""Does a management-interface poller running BESIDE the kernels change their times?  The driver's bench record lists ~30 `smi.<epoch>.json`
files written every ~5 s during its run (BENCH_r04.json: pulled_files, gpu_busy.samples) -- something the builder's gpurun calls do not
have -- and on the driver's box the 16-bit MFMA kernels (inference AND training) ran at a HIGHER granted clock with MORE busy cycles,
i.e. stalled.  Phases: quiet / one poller style in a tight loop / the same every 5 s / quiet.  Per phase: the three inference kernels
(65536 x 192 launch) and the bf16x3 / f16x3 training kernels at 262144 points (bench.train_roofline)."""
import argparse, os, shutil, signal, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
from nerf import ops
dev = torch.device("cuda:0")
ro, rd = nerf.get_ray_bundle(512, 512, bench.INTRINSICS, bench.frame_pose(0).to(dev))
n_rays, S = 65536, 192
ro_, rd_ = ro.view(-1, 3)[:n_rays].contiguous(), rd.view(-1, 3)[:n_rays].contiguous()
m = bench.synth_params(1, dev)
expr, lat = torch.randn(76, device=dev) * 0.5, torch.randn(32, device=dev) * 0.1
z = torch.sort(torch.rand((n_rays, S), device=dev) * 0.6 + 0.2, dim=-1)[0].contiguous()
hw = m.hip_weights()
cond = ops.paper_condition(hw.get(), expr, lat, 0.2, 0.8)
INF = (("f32", lambda: ops.paper_mlp_fwd(hw.get(), cond, ro_, rd_, z)),
       ("bf16x3", lambda: ops.paper_mlp_fwd_bf16(hw.get_bf16(), cond, ro_, rd_, z)),
       ("f16x3", lambda: ops.paper_mlp_fwd_f16(hw.get_f16(), cond, ro_, rd_, z)))
mt = bench.synth_params(1, dev).train()

def measure(tag, train=True):
    out = []
    for name, fn in INF:
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(6 if name == "f32" else 12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        out.append(f"{name} {ts[0]:.2f}/{ts[len(ts) // 2]:.2f}/{ts[-1]:.2f}")
    if train:
        for prec in ("bf16x3", "f16x3"):
            nerf.set_mlp_precision(prec)
            r = bench.train_roofline(argparse.Namespace(precision=prec, family="paper"), mt, dev, 2048)
            ks = r["kernels"]
            out.append(f"train {prec} fwd/chain/dw {ks[0]['avg_launch_ms']:.3f}/{ks[1]['avg_launch_ms']:.3f}/{ks[2]['avg_launch_ms']:.3f}")
        nerf.set_mlp_precision("f32")
    print(f"[{tag:34s}] inference ms min/med/max: " + " | ".join(out), flush=True)

def poller(cmd, period):
    sh = f"while true; do {cmd} > /dev/null 2>&1; " + (f"sleep {period}; " if period else "") + "done"
    return subprocess.Popen(["bash", "-c", sh], preexec_fn=os.setsid)

def stop(p):
    os.killpg(os.getpgid(p.pid), signal.SIGTERM)
    p.wait()

STYLES = [("rocm-smi --showuse --json", "rocm-smi --showuse --json"),
          ("rocm-smi -a --json", "rocm-smi -a --json"),
          ("rocm-smi power/clocks/temp/mem", "rocm-smi --showpower --showclocks --showtemp --showmemuse --showmeminfo vram --json"),
          ("amd-smi metric --json", "amd-smi metric --json"),
          ("amd-smi monitor (one shot)", "amd-smi monitor -p -t -u -m -v"),
          ("cat gpu_metrics (sysfs)", "cat /sys/class/drm/card*/device/gpu_metrics"),
          ("cat gpu_busy_percent (sysfs)", "cat /sys/class/drm/card*/device/gpu_busy_percent /sys/class/drm/card*/device/mem_busy_percent")]
measure("quiet (start)")
for label, cmd in STYLES:
    exe = cmd.split()[0]
    if exe not in ("cat",) and shutil.which(exe) is None:
        print(f"[{label}] not installed", flush=True)
        continue
    t0 = time.time(); subprocess.run(["bash", "-c", cmd + " > /dev/null 2>&1"]); one = time.time() - t0
    p = poller(cmd, 0)
    time.sleep(1.0)
    measure(f"{label}: tight loop ({one:.2f} s/call)")
    stop(p)
    p = poller(cmd, 5)
    time.sleep(0.5)
    measure(f"{label}: every 5 s", train=False)
    stop(p)
measure("quiet (end)")
