# round 4, call 33: full GPU suite, the bench line as the driver runs it, and the kernel trace of the same command (profiles/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c33; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v Warning | tail -25 > $O/pytest_gpu.txt; grep -n "passed\|failed\|Error" $O/pytest_gpu.txt | tail -5
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench_time.txt; cat $O/bench_time.txt; tail -c 2500 $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --train-steps 10 > $GRAFT_REPO_ROOT/$O/bench_under_trace.json 2> $GRAFT_REPO_ROOT/$O/bench_under_trace.err
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/kt -name "*.db" | head -1) > $GRAFT_REPO_ROOT/$O/bench_kernel_stats.md 2>&1
head -30 $GRAFT_REPO_ROOT/$O/bench_kernel_stats.md | cut -c1-140
