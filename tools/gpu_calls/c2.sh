# GPU call 2 (round 3): software-pipelined K loops (all exact-f32 kernels), balanced dW slices: A/B, eval-kernel timing, full GPU test suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
timeout 300 python tools/ab_train_f32.py --odd --json gpurun_out/c2/ab.json > gpurun_out/c2/ab.log 2>&1; echo "ab rc=$?"
grep -v amdgpu.ids gpurun_out/c2/ab.log | tail -6
TIME_MLP_SKIP_SPLIT=1 timeout 300 python tools/time_mlp.py > gpurun_out/c2/time_mlp.log 2>&1; echo "time_mlp rc=$?"
grep -v amdgpu.ids gpurun_out/c2/time_mlp.log | tail -8
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/c2/t_gpu.log 2>&1; echo "gpu tests rc=$?"
tail -6 gpurun_out/c2/t_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ab -o ab -- python $GRAFT_REPO_ROOT/tools/ab_train_f32.py --iters 5 > $GRAFT_REPO_ROOT/gpurun_out/c2/prof.log 2>&1; echo "prof rc=$?"
DB=$(find /tmp/prof_ab -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $DB > $GRAFT_REPO_ROOT/gpurun_out/c2/ab_kernel_stats.md 2>&1
head -14 $GRAFT_REPO_ROOT/gpurun_out/c2/ab_kernel_stats.md | cut -c1-150
