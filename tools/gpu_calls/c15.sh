# GPU call 15 (round 3): the layer-streamed exact-f32 inference kernel (nf_mlp_stream.h) against the round-2 form on one box:
# bit-level fingerprints of the outputs (must be equal) and HIP-event timings of the fine / coarse launches, 5 variants
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c15
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in r2 "" pipe persist pipe_persist r2 ""; do
  lib=$L/libnerface_hip${v:+_$v}.so
  echo "== variant ${v:-default} ($lib)"
  NERFACE_HIP_LIB=$lib TIME_MLP_ONLY_F32=1 TIME_MLP_HASH=1 timeout 300 python tools/time_mlp.py 2>&1 | grep -v Warning
done | tee gpurun_out/c15/variants.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "paper_mlp_fwd or render_rays or pipeline" 2>&1 | tail -3
