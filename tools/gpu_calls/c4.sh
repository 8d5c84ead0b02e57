# GPU call 4 (round 3): full GPU test suite (new tests: RCCL world 1, 8 ranks gloo, f16 guard in training, stochastic full frame, ties), then bench.py
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/c4/t_gpu.log 2>&1; echo "gpu tests rc=$?"
tail -30 gpurun_out/c4/t_gpu.log
timeout 900 python bench.py > gpurun_out/c4/bench.json 2> gpurun_out/c4/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/c4/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/c4/bench.json') if l.startswith('{')][-1])
print('value',d['value'],'ms',d['ms_per_step'],'roofline',{k:d['roofline'].get(k) for k in ('frac','frac_executed','avg_launch_ms','traffic')})
for p,t in d['train'].items():
    if p=='workload': continue
    print(p, 'ms/iter', t['ms_per_iter'], [(k['kernel'].split()[0], round(k['avg_launch_ms'],3), k['bound'], round(k['frac'],3), k['traffic']) for k in t['roofline']['kernels']])
print('eager', d.get('eager_rocm')); print('cpu', {k:v for k,v in d.get('cpu_baseline',{}).items() if k in ('value','cores','host_cores','threads')})
PY
