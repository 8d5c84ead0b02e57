# GPU call 16 (round 3): timing ablations of the layer-streamed f32 inference kernel (results of the ablated builds are invalid by construction)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c16
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in "" noload nolds noboth nope noall ""; do
  lib=$L/libnerface_hip${v:+_$v}.so
  echo "== variant ${v:-default}"
  NERFACE_HIP_LIB=$lib TIME_MLP_ONLY_F32=1 timeout 300 python tools/time_mlp.py 2>&1 | grep "f32 "
done | tee gpurun_out/c16/variants.txt
