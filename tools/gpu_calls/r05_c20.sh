# round 5, call 20: the new multi-frame gate test of f16x2
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c20; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_f16x2.py -m gpu -q -s -k "gate_over_frames" 2>&1 | grep -v Warning | tail -14 | tee $O/pytest.txt
