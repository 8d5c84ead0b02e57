# round 5, call 13: can the driver-box signature (16-bit MFMA kernels slower at a HIGHER clock) be produced by a power-management setting?
# the same kernels under perf level auto (the pool's default), forced high, and a lowered power cap; settings restored at the end
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c13; mkdir -p $O
( timeout 200 python tools/perf_level_probe.py auto
  rocm-smi --setperflevel high 2>&1 | grep -v "^$" | tail -2
  timeout 200 python tools/perf_level_probe.py high
  rocm-smi --setperflevel auto 2>&1 | grep -v "^$" | tail -2
  rocm-smi --setpoweroverdrive 1000 --autorespond y 2>&1 | grep -v "^$" | tail -3
  timeout 200 python tools/perf_level_probe.py cap1000
  rocm-smi --setperflevel high 2>&1 | grep -v "^$" | tail -1
  timeout 200 python tools/perf_level_probe.py cap1000+high
  rocm-smi --setperflevel auto 2>&1 | grep -v "^$" | tail -1
  rocm-smi --resetpoweroverdrive --autorespond y 2>&1 | grep -v "^$" | tail -2
  timeout 200 python tools/perf_level_probe.py restored ) 2>&1 | grep -v "Warning\|amdgpu.ids" > $O/perf_level_probe.txt
cat $O/perf_level_probe.txt
