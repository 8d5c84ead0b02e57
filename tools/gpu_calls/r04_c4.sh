# round 4, call 4: training forwards save the dW kernel's fragment stream (transposing MFMAs), dW DMAs streamed tiles:
# parity suites, the new resample/merge kernel's test, then kernel times against round 3 and against the deferred-LDS-save build (c3)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c4; mkdir -p $O
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_lcode.py tests/test_gpu_bf16.py tests/test_gpu_f16.py tests/test_gpu_kernels.py -q -m gpu 2>&1 | tail -40 > $O/pytest_split.txt; grep -n "passed\|failed\|Error" $O/pytest_split.txt | tail -8
for v in _r03 _c3 "" _c3 ""; do echo "== lib${v:-_new}"; NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 300 python tools/time_train_split.py bf16x3 f16x3 lcode 2>&1 | grep "paper\|lcode"; done > $O/train_ab.txt; cat $O/train_ab.txt
