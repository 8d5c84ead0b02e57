# GPU call 20 (round 3): timing ablations of the final f32 inference kernel (pipelined buffer-load K loop, persistent grid); ablated builds give invalid results
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c20
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in "" noload nolds noboth nope noall ""; do
  lib=$L/libnerface_hip${v:+_$v}.so
  echo "== variant ${v:-default}"
  NERFACE_HIP_LIB=$lib TIME_MLP_ONLY_F32=1 timeout 300 python tools/time_mlp.py 2>&1 | grep "f32 "
done | tee gpurun_out/c20/variants.txt
