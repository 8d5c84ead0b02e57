# round 4, call 17: k_tiny_mlp_fwd and k_paper_mlp_fwd_encoded on the streamed K loops: fingerprints against the previous library, their tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c17; mkdir -p $O
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in _prev "" _prev ""; do echo "== lib${v:-_new}"; NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 300 python tools/stream_port_check.py 2>&1 | grep "tiny\|encoded\|Error" | tail -12; done > $O/ports_ab.txt; cat $O/ports_ab.txt
timeout 900 python -m pytest tests/test_gpu_tiny.py tests/test_gpu_kernels.py -q -m gpu --tb=short 2>&1 | grep -v Warning | tail -8 > $O/pytest.txt; grep -n "passed\|failed\|Error\|assert" $O/pytest.txt | tail -6
