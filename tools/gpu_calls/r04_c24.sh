# round 4, call 24: the fused training loss: parity test, launcher tests, iteration time against the torch expression
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_launchers.py -q -m gpu --tb=short -k "training_loss or launcher or train or resume or sharded" 2>&1 | grep -v Warning | tail -25 > $O/pytest.txt; tail -12 $O/pytest.txt
timeout 600 python tools/train_step_ab.py 2>&1 | grep "loss:\|Error\|error" | tee $O/train_step_ab.txt
