# GPU call 24 (round 3): buffer-form LDS DMA in the weight-gradient kernel (k_dw_gemm_lds): A/B + backward parity (both model families)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c24
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in dwglobal "" dwglobal ""; do
  echo "== variant ${v:-default}"
  NERFACE_HIP_LIB=$L/libnerface_hip${v:+_$v}.so timeout 300 python tools/time_train_f32.py 10 2>&1 | grep "ms"
done | tee gpurun_out/c24/variants.txt
timeout 600 python -m pytest tests/test_gpu_backward.py tests/test_gpu_lcode.py -q -m gpu -x 2>&1 | tail -3
