# GPU call 1 (round 3): A/B of the exact-f32 training kernels, backward tests, kernel trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
timeout 300 python tools/ab_train_f32.py --odd --json gpurun_out/c1/ab.json > gpurun_out/c1/ab.log 2>&1; echo "ab rc=$?"
tail -8 gpurun_out/c1/ab.log
timeout 600 python -m pytest tests/test_gpu_backward.py -x -q -m gpu > gpurun_out/c1/t_bwd.log 2>&1; echo "bwd tests rc=$?"
tail -15 gpurun_out/c1/t_bwd.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o ab -- python $GRAFT_REPO_ROOT/tools/ab_train_f32.py --iters 5 > $GRAFT_REPO_ROOT/gpurun_out/c1/prof.log 2>&1; echo "prof rc=$?"
DB=$(find /tmp/prof_ab -name "*.db" | head -1); echo $DB
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $DB > $GRAFT_REPO_ROOT/gpurun_out/c1/ab_kernel_stats.md 2>&1
head -30 $GRAFT_REPO_ROOT/gpurun_out/c1/ab_kernel_stats.md
