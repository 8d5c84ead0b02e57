# round 5, call 10: numerical model of "one fp16 product per weight" with first-order corrections (oracle/split_emulation.py, fp64 on the device)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c10; mkdir -p $O
timeout 1200 python -m oracle.split_emulation > $O/split_emulation.txt 2>&1; grep -v Warning $O/split_emulation.txt | tail -14
