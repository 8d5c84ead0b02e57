# round 5, call 8: "f16x2" shipped as the fourth arithmetic (ABI 3): full GPU suite (new: test_gpu_f16x2.py, whole frames in four arithmetics,
# second family whole frame), then a default bench run with the split_f16x2 leg
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c8; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v Warning > $O/pytest_gpu_full.txt; tail -6 $O/pytest_gpu_full.txt; grep -a "f16x2" $O/pytest_gpu_full.txt | cut -c1-260 | head -40
timeout 900 python3 bench.py > $O/bench.out 2> $O/bench.err; echo "bench rc $?"; tail -n 1 $O/bench.out | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(len(json.dumps(d))); print({k:v for k,v in d['summary'].items() if 'f16' in k or 'eager' in k or k in ('value_rays_s','roofline_frac','roofline_avg_launch_ms','train_ms_per_iter_f32','train_ms_per_iter_bf16x3')})"
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
tail -5 $O/bench.err
