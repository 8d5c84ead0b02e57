# round 5, call 9: re-gated raw-output tests of f16x2; A/B of the x2 kernel with the A fragments of 8 output tiles at a time (variant library,
# alternating processes)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c9; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_f16x2.py -m gpu -q -s 2>&1 | grep -v Warning | tail -14 > $O/pytest_f16x2.txt; cat $O/pytest_f16x2.txt | cut -c1-300
for rep in 1 2 3; do
  TIME_MLP_ONLY=f16x2 timeout 200 python tools/time_mlp.py 2>&1 | grep f16x2 | sed 's/^/tg4 /' | tee -a $O/tg.txt
  NERFACE_HIP_LIB=$PWD/4d-facial-avatars_amd/lib/libnerface_hip_xtg8.so TIME_MLP_ONLY=f16x2 timeout 200 python tools/time_mlp.py 2>&1 | grep f16x2 | sed 's/^/tg8 /' | tee -a $O/tg.txt
done
