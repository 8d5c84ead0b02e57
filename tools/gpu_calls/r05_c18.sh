# round 5, call 18: numerical model of MIXED arithmetics (22-bit weights only where the density is made, 11-bit + bias correction elsewhere)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c18; mkdir -p $O
timeout 1200 python -m oracle.split_emulation mixes > $O/split_emulation_mixes.txt 2>&1; grep -v Warning $O/split_emulation_mixes.txt | tail -20
