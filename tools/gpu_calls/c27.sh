# GPU call 27 (round 3, last 48 s of the budget): first run of -DNF_FWD_PREFETCH_IN=1 -- fingerprints against the default build, fine-launch time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c27
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in "" prefetch; do
  echo "== variant ${v:-default}"
  NERFACE_HIP_LIB=$L/libnerface_hip${v:+_$v}.so TIME_MLP_ONLY_F32=1 TIME_MLP_HASH=1 timeout 20 python tools/time_mlp.py 2>&1 | grep "f32 " | cut -c1-75
done | tee gpurun_out/c27/prefetch.txt
