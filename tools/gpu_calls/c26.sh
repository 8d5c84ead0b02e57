# GPU call 26 (round 3): kernel trace of the headline bench (bench.py --no-extras --no-cpu-baseline) with the duration-cluster table: the persistent
# inference kernel launches one grid for every size, so its coarse / fine launches are told apart by duration
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/c26
timeout 50 rocprofv3 --kernel-trace --stats -d /tmp/prof_h -o h -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/c26/bench_line.json 2> /dev/null; echo "rc=$?"
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/prof_h -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/c26/headline_kernel_stats.md 2>&1
grep -A8 "duration clusters" $GRAFT_REPO_ROOT/gpurun_out/c26/headline_kernel_stats.md | cut -c1-120
