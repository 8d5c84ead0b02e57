# round 5, call 15: the launchers with --precision f16x2 (both model families)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_launchers.py -m gpu -q 2>&1 | grep -v Warning | tail -6 > $O/pytest_launchers.txt; cat $O/pytest_launchers.txt
