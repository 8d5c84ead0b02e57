# round 4, call 5: failing cases of call 4 with their messages; box write-rate probe beside the kernel times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_kernels.py -q -m gpu -x -s --tb=short -k "training_kernels_vs_fp64_at_training_size or resample_merge" 2>&1 | grep -v Warning | tail -60 > $O/pytest_fail.txt; tail -45 $O/pytest_fail.txt
timeout 300 python tools/time_train_split.py bf16x3 f16x3 2>&1 | grep "paper\|lcode\|box" | tee $O/train_new.txt
