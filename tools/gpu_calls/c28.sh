# GPU call 28 (round 3, the last seconds of the budget): first run of -DNF_LCODE_STREAM=1 -- the second family's f32 inference parity tests on the variant library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c28
NERFACE_HIP_LIB=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib/libnerface_hip_lcode_stream.so timeout 22 python -m pytest tests/test_gpu_lcode.py -q -m gpu -x -k "eval_against_golden or mlp_vs_fp64_oracle or bwd_vs_fp64_oracle and f32" 2>&1 | tail -3 | tee gpurun_out/c28/lcode.txt
