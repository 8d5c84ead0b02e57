# round 5, call 22: final state -- full GPU suite (-x, as the driver runs it), smoke(), the driver's exact bench command
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c22; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -v Warning | tail -6 > $O/pytest_gpu.txt; grep -n "passed\|failed\|Error" $O/pytest_gpu.txt | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err ) 2> $O/bench_time.txt; grep real $O/bench_time.txt
tail -n 1 $O/bench.out | python3 -c "import sys,json; t=sys.stdin.read(); d=json.loads(t); s=d['summary']; print(len(t), d['value'], d['roofline']['frac'], d['cpu_baseline']['kind'], {k: s.get(k) for k in ('split_f16x2_rays_s','split_f16x2_over_eager','train_ms_per_iter_f32','train_ms_per_iter_f16x3','train_ms_per_iter_bf16x3','power_w_train_fwd_bf16x3')})"
