# round 4, call 6: fused training epilogue + interleaved transposing MFMAs + cheaper mask bits: parity suites, kernel times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_lcode.py tests/test_gpu_bf16.py tests/test_gpu_f16.py tests/test_gpu_kernels.py -q -m gpu --tb=short 2>&1 | grep -v Warning | tail -40 > $O/pytest_split.txt; grep -n "passed\|failed\|Error\|assert" $O/pytest_split.txt | tail -12
timeout 300 python tools/time_train_split.py bf16x3 f16x3 lcode 2>&1 | grep "paper\|lcode\|box" | tee $O/train_new.txt
