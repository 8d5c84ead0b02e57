# round 4, call 14: streamed f32 chains (both families) as the only form: lcode fingerprints against the previous library, full backward suites
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c14; mkdir -p $O
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in _prev "" _prev ""; do echo "== lib${v:-_new}"; NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 300 python tools/lcode_chain_check.py 2>&1 | grep "lcode"; done > $O/lcode_chain_ab.txt; cat $O/lcode_chain_ab.txt
timeout 300 python tools/chain_check.py 2>&1 | grep "sha1\|f32 @" | tee $O/paper_chain.txt
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_lcode.py tests/test_gpu_launchers.py -q -m gpu --tb=short 2>&1 | grep -v Warning | tail -12 > $O/pytest.txt; grep -n "passed\|failed\|Error\|assert" $O/pytest.txt | tail -6
