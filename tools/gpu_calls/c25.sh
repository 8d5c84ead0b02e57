# GPU call 25 (round 3, final): full GPU test suite, smoke(), bench.py (-> profiles/r03_bench_line.json), kernel traces of the bench and of the f32 training iteration
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c25
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/c25/t_gpu.log 2>&1; echo "gpu tests rc=$?"
tail -4 gpurun_out/c25/t_gpu.log | cut -c1-300
timeout 200 python -c "import __graft_entry__ as G; G.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/c25/bench.json 2> gpurun_out/c25/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c25/bench.json') if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step'],'roofline',{k:d['roofline'].get(k) for k in ('frac','frac_executed','avg_launch_ms','traffic')})
print({k:v for k,v in d['config']['device'].items() if 'hbm' in k})
for p,t in d['train'].items():
    if p=='workload': continue
    print(p, 'ms/iter', t['ms_per_iter'], [(k['kernel'].split()[0], round(k['avg_launch_ms'],3), k['bound'], round(k['frac'],3)) for k in t['roofline']['kernels']])
PY
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --no-extras > $GRAFT_REPO_ROOT/gpurun_out/c25/bench_profiled.json 2> /dev/null; echo "prof bench rc=$?"
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/prof_b -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/c25/bench_kernel_stats.md 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 40 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/c25/train_line.json 2> /dev/null; echo "prof train rc=$?"
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/prof_tr -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/c25/train_iter_kernel_stats.md 2>&1
grep -A8 "per (kernel, grid size)" $GRAFT_REPO_ROOT/gpurun_out/c25/train_iter_kernel_stats.md | cut -c1-110
