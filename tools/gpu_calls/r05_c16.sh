# round 5, call 16: headline kernel with the block inputs fetched one block ahead (B = this tree) against the kernel before the change
# (A = lib/libnerface_hip_xold.so): output hashes at four launch shapes, fine / coarse launch times, alternating processes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c16; mkdir -p $O
for rep in 1 2 3; do
  NERFACE_HIP_LIB=$PWD/4d-facial-avatars_amd/lib/libnerface_hip_xold.so TIME_MLP_ONLY=f32 TIME_MLP_HASH=1 timeout 200 python tools/time_mlp.py 2>&1 | grep "f32" | sed 's/^/A /' | tee -a $O/ab.txt
  TIME_MLP_ONLY=f32 TIME_MLP_HASH=1 timeout 200 python tools/time_mlp.py 2>&1 | grep "f32" | sed 's/^/B /' | tee -a $O/ab.txt
done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | grep -v Warning | tail -3 | tee $O/pytest.txt
