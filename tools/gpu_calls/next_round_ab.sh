# First GPU run of the two switches left unmeasured at the end of round 3 (python tools/build_next_variants.py first):
#  * prefetch: measured once in call 27 (bit-identical, no gain); kept here as the A/B template
#  * lcode_stream: the second family's parity tests on the variant library, then its eval timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/next_ab
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in "" prefetch "" prefetch; do
  echo "== variant ${v:-default}"
  NERFACE_HIP_LIB=$L/libnerface_hip${v:+_$v}.so TIME_MLP_ONLY_F32=1 TIME_MLP_HASH=1 timeout 300 python tools/time_mlp.py 2>&1 | grep "f32 " | cut -c1-75
done | tee gpurun_out/next_ab/prefetch.txt
NERFACE_HIP_LIB=$L/libnerface_hip_lcode_stream.so timeout 600 python -m pytest tests/test_gpu_lcode.py -q -m gpu -x 2>&1 | tail -3 | tee gpurun_out/next_ab/lcode_parity.txt
