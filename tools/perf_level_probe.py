"""Launch time, socket power and engine clock of the inference kernels and the split training forward under the box's CURRENT power-management
settings (bench.power_probe: each kernel back to back for 1.5 s, sysfs sampled every 20 ms).  Run in round 5 (call 13; now: bash tools/gpu_call.sh TAG py tools/perf_level_probe.py) under
`auto`, a forced `high` performance level and a lowered power cap: which setting produces the driver-box signature of rounds 3-4 (16-bit MFMA
kernels slower at a HIGHER reported clock)?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "4d-facial-avatars_amd"))
import torch, bench, nerf
from nerf import ops
dev = torch.device("cuda:0")
m = bench.synth_params(1, dev)
hw = m.hip_weights()
pk, pk_h, pk_b = hw.get(), hw.get_f16(), hw.get_bf16()
cond = ops.paper_condition(pk, torch.randn(76, device=dev) * 0.5, torch.randn(32, device=dev) * 0.1, 0.2, 0.8)
ro, rd = nerf.get_ray_bundle(512, 512, bench.INTRINSICS, bench.frame_pose(0).to(dev))
ro, rd = ro.view(-1, 3)[:65536].contiguous(), rd.view(-1, 3)[:65536].contiguous()
z = torch.sort(torch.rand((65536, 192), device=dev) * 0.6 + 0.2, dim=-1)[0].contiguous()
zt = z[:2048, :128].contiguous()
out = bench.power_probe(dev, (("f32", lambda: ops.paper_mlp_fwd(pk, cond, ro, rd, z)),
                              ("f16x3", lambda: ops.paper_mlp_fwd_f16(pk_h, cond, ro, rd, z)),
                              ("f16x2", lambda: ops.paper_mlp_fwd_f16x2(pk_h, cond, ro, rd, z)),
                              ("train_fwd_bf16x3", lambda: ops.paper_mlp_fwd_train(pk, cond, ro[:2048], rd[:2048], zt, packed_b=pk_b)),
                              ("train_fwd_f32", lambda: ops.paper_mlp_fwd_train(pk, cond, ro[:2048], rd[:2048], zt))))
st = out.get("static", {})
print(f"[{sys.argv[1] if len(sys.argv) > 1 else ''}] perf_level {st.get('perf_level')}, cap {st.get('power_cap_w')} W, sclk table {str(st.get('pp_dpm_sclk')).replace(chr(10), ' / ')}")
for k, o in out.items():
    if k != "static":
        print(f"    {k:18s} {o['launch_ms']:8.3f} ms  {o['power_w'] or 0:7.1f} W  sclk {o['sclk_mhz_hwmon'] or 0:7.1f} MHz  fclk {o['fclk_mhz_dpm'] or 0:6.0f}")
