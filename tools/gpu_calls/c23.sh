# GPU call 23 (round 3): details of the failing backward parity case
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_backward.py -q -m gpu -x 2>&1 | grep -v Warning | tail -8; timeout 200 python tools/time_train_f32.py 10 2>&1 | grep ms
