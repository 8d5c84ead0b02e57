# GPU call 18 (round 3): layer-streamed f32 inference kernel, pipelined K loop + buffer loads (weights and bias) as the default;
# persistent grid and tail lag variants; fingerprints must equal the round-2 kernel's
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c18
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in r2 "" persist lag1 lag3 lag3_persist ""; do
  lib=$L/libnerface_hip${v:+_$v}.so
  echo "== variant ${v:-default}"
  NERFACE_HIP_LIB=$lib TIME_MLP_ONLY_F32=1 TIME_MLP_HASH=1 timeout 300 python tools/time_mlp.py 2>&1 | grep "f32 " | cut -c1-75
done | tee gpurun_out/c18/variants.txt
