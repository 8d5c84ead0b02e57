# round 4, call 16: the second family's exact-f32 training forward on the streamed K loops: fingerprints and time against the previous library, suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c16; mkdir -p $O
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in _prev "" _prev ""; do echo "== lib${v:-_new}"; NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 300 python tools/lcode_train_check.py 2>&1 | grep "lcode\|Error" | tail -6; done > $O/lcode_save_ab.txt; cat $O/lcode_save_ab.txt
timeout 900 python -m pytest tests/test_gpu_lcode.py -q -m gpu --tb=short 2>&1 | grep -v Warning | tail -6 > $O/pytest.txt; grep -n "passed\|failed\|Error\|assert" $O/pytest.txt | tail -4
