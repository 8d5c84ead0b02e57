# round 5, call 11: full GPU suite, smoke(), the bench line as the driver runs it, and the kernel trace of the same command (profiles/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c11; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v Warning | tail -25 > $O/pytest_gpu.txt; grep -n "passed\|failed\|Error" $O/pytest_gpu.txt | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err ) 2> $O/bench_time.txt; grep real $O/bench_time.txt; tail -n 1 $O/bench.out | cut -c1-5500
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --train-steps 10 > $GRAFT_REPO_ROOT/$O/bench_under_trace.json 2> $GRAFT_REPO_ROOT/$O/bench_under_trace.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/kt -name "*.db" | head -1) > $GRAFT_REPO_ROOT/$O/bench_kernel_stats.md 2>&1
head -12 $GRAFT_REPO_ROOT/$O/bench_kernel_stats.md | cut -c1-140
