# GPU call 19 (round 3): buffer-load weights in the exact-f32 TRAINING kernels' K loops (nf_mma_from_lds_side): A/B per kernel + parity tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c19
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in tglobal "" tglobal ""; do
  echo "== variant ${v:-default}"
  NERFACE_HIP_LIB=$L/libnerface_hip${v:+_$v}.so timeout 300 python tools/time_train_f32.py 10 2>&1 | grep -v Warn | grep "ms"
done | tee gpurun_out/c19/variants.txt
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_lcode.py tests/test_gpu_kernels.py -q -m gpu -x 2>&1 | tail -3
