# GPU call 21 (round 3): layer-streamed exact-f32 TRAINING forward (ReLU, mask bits and the copy to `saved` at the consumer side of the slab):
# per-kernel times, parity tests, and the inference kernel's fingerprint after the refactor (must still equal the round-2 kernel's)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c21
timeout 300 python tools/time_train_f32.py 10 2>&1 | grep "ms" | tee gpurun_out/c21/train.txt
TIME_MLP_ONLY_F32=1 TIME_MLP_HASH=1 timeout 300 python tools/time_mlp.py 2>&1 | grep "f32 " | cut -c1-75 | tee gpurun_out/c21/eval.txt
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -5
