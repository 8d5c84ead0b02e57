# GPU call 5 (round 3): same-box variants of the f32 training kernels (dW stagger, ablations of the forward's masks / copies)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in "" _nostagger _sleep2 _sleep8 _nomask _nocopy ""; do
  echo "== variant '$v'"
  NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 120 python tools/time_train_f32.py 10 2>&1 | grep -v amdgpu.ids
done > gpurun_out/c5/variants.log 2>&1
cat gpurun_out/c5/variants.log
timeout 200 python tools/ab_train_f32.py --odd 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tail -4
