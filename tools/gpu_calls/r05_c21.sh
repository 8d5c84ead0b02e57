# round 5, call 21: launcher tests incl. eval_sharded --verify-gate
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_launchers.py -m gpu -q -s -k "roundtrip" 2>&1 | grep -v Warning | grep "gate\|passed\|failed\|Error\|assert" | cut -c1-400 | tee $O/pytest.txt
