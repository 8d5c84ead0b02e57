# GPU call 14 (round 3): bench.py on another box (the box of call 13 ran every store-heavy kernel 2x slower) + its kernel trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c14
timeout 900 python bench.py > gpurun_out/c14/bench.json 2> gpurun_out/c14/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c14/bench.json') if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step'],'roofline',{k:d['roofline'].get(k) for k in ('frac','frac_executed','avg_launch_ms','traffic')})
print({k:v for k,v in d['config']['device'].items() if 'hbm' in k})
for p,t in d['train'].items():
    if p=='workload': continue
    print(p, 'ms/iter', t['ms_per_iter'], [(k['kernel'].split()[0], round(k['avg_launch_ms'],3), k['bound'], round(k['frac'],3)) for k in t['roofline']['kernels']])
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/c14/bench_profiled.json 2> /dev/null; echo "prof bench rc=$?"
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/prof_b -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/c14/bench_kernel_stats.md 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 40 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/c14/train_line.json 2> /dev/null; echo "prof train rc=$?"
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/prof_tr -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/c14/train_iter_kernel_stats.md 2>&1
grep -A8 "per (kernel, grid size)" $GRAFT_REPO_ROOT/gpurun_out/c14/train_iter_kernel_stats.md | cut -c1-110
