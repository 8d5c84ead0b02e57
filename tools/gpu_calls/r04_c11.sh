# round 4, call 11: the weight-stationary split inference kernel: fingerprints against the shipped kernel, launch times, parity suites
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c11; mkdir -p $O
for ws in 0 1 0 1; do echo "== NERFACE_SPLIT_WS=$ws"; NERFACE_SPLIT_WS=$ws timeout 300 python tools/ws_check.py 2>&1 | grep "x3"; done > $O/ws_ab.txt; cat $O/ws_ab.txt
NERFACE_SPLIT_WS=1 timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_f16.py tests/test_gpu_e2e.py -q -m gpu --tb=short 2>&1 | grep -v Warning | tail -15 > $O/pytest_ws.txt; grep -n "passed\|failed\|Error\|assert" $O/pytest_ws.txt | tail
