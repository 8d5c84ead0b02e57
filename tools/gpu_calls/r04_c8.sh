# round 4, call 8: kernel-trace durations (ground truth) of the split training kernels, new library
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_c8; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o p -- python $GRAFT_REPO_ROOT/tools/pmc_train_launch.py bf16x3 f16x3 f32 > $O/kt1.log 2>&1; echo "kt1 rc=$?"
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $(find /tmp/kt1 -name "*.db" | head -1) > $O/train_kernel_stats.md 2>&1
grep "k_paper\|k_dw\|k_grad" $O/train_kernel_stats.md | head -30
