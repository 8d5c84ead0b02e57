# GPU call 17 (round 3): K-loop forms of the layer-streamed f32 inference kernel: compiler-scheduled / one load per tile / burst at the top
# of the chunk, global vs buffer loads; fingerprints must equal the round-2 kernel's
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c17
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in r2 "" wbuf pipe_wbuf burst burst_wbuf burst_wbuf_persist ""; do
  lib=$L/libnerface_hip${v:+_$v}.so
  echo "== variant ${v:-default}"
  NERFACE_HIP_LIB=$lib TIME_MLP_ONLY_F32=1 TIME_MLP_HASH=1 timeout 300 python tools/time_mlp.py 2>&1 | grep "f32 " | cut -c1-75
done | tee gpurun_out/c17/variants.txt
