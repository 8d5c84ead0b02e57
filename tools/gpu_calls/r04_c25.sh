# round 4, call 25: ray selection one iteration ahead on a side stream: launcher tests, iteration time against the in-line draw
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c25; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_launchers.py tests/test_gpu_dropin_scripts.py -q -m gpu --tb=short 2>&1 | grep -v Warning | tail -25 > $O/pytest.txt; tail -8 $O/pytest.txt
timeout 600 python tools/train_step_ab.py 2>&1 | grep "per iteration\|Error\|error" | tee $O/train_step_ab.txt
