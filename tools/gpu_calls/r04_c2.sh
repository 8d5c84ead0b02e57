# round 4, call 2: the split training kernels with deferred saves (two accumulator sets) + buffer-form weight DMA:
# parity suites on the new library, then A/B kernel times against the round-3 library (same box, same process order), inference too
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c2; mkdir -p $O
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_lcode.py tests/test_gpu_bf16.py tests/test_gpu_f16.py -q -m gpu 2>&1 | tail -25 > $O/pytest_split.txt; tail -5 $O/pytest_split.txt
for v in _r03 "" _r03 ""; do echo "== lib${v:-_new}"; NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 300 python tools/time_train_split.py bf16x3 f16x3 lcode 2>&1 | grep "paper\|lcode"; done > $O/train_ab.txt; cat $O/train_ab.txt
for v in _r03 "" _r03 ""; do echo "== lib${v:-_new}"; NERFACE_HIP_LIB=$L/libnerface_hip$v.so TIME_MLP_SKIP_F32=1 timeout 300 python tools/time_mlp.py 2>&1 | grep "x3"; done > $O/infer_ab.txt; cat $O/infer_ab.txt
timeout 900 python -m pytest tests/test_gpu_launchers.py -q -m gpu -x -k "bench_two_ranks" 2>&1 | tail -60 > $O/pytest_bench2.txt; grep -n "Error\|error\|assert" $O/pytest_bench2.txt | head -20
