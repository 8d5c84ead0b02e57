# round 5, call 19: north_star's gate over 16 frames of the bench scene in the three split arithmetics (tools/frame_gate_sweep.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c19; mkdir -p $O
timeout 600 python tools/frame_gate_sweep.py 16 2>&1 | grep "^frame\|^worst" | tee $O/frame_gate_sweep.txt
