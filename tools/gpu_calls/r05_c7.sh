# round 5, call 7: the whole-frame north_star gates (|dPSNR| <= 1e-4 dB against the fp64 oracle, deterministic and perturbed, hard and soft
# density head) with the "no W_hi x_lo" variant of the split kernels in place of f16x3 / bf16x3 (lib/libnerface_hip_xall_no_xlo.so)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c7; mkdir -p $O
NERFACE_HIP_LIB=$PWD/4d-facial-avatars_amd/lib/libnerface_hip_xall_no_xlo.so timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "full_frame_512_vs or full_frame_512_stochastic" 2>&1 | grep -v Warning > $O/full_frame_no_xlo.txt
grep -a "full frame\|passed\|failed\|assert" $O/full_frame_no_xlo.txt | cut -c1-220
NERFACE_HIP_LIB=$PWD/4d-facial-avatars_amd/lib/libnerface_hip_xall_main_only.so timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -s -k "full_frame_512_vs" 2>&1 | grep -v Warning > $O/full_frame_main_only.txt
grep -a "full frame\|passed\|failed" $O/full_frame_main_only.txt | cut -c1-220
