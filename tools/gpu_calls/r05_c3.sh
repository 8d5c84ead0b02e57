# round 5, call 3: the driver's bench record carries ~30 smi.<epoch>.json files written every ~5 s during ITS run; the builder's calls have
# no such poller.  Do management-interface queries beside the kernels reproduce the driver-box signature (16-bit MFMA kernels slower at a
# HIGHER granted clock)?  tools/smi_poll_probe.py: quiet / poller in a tight loop / every 5 s, several poller styles.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c3; mkdir -p $O
which rocm-smi amd-smi > $O/which.txt 2>&1
timeout 1200 python tools/smi_poll_probe.py > $O/smi_poll_probe.txt 2>&1; cat $O/smi_poll_probe.txt | grep -v Warning | tail -30
