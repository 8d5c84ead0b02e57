# round 4, call 3: what bounds the deferred saves?  Timing ablations of the store path (variant libraries, results invalid except _new/_nont),
# then the full GPU suite on the cleaned default library (streamed lcode inference promoted, experiment switches removed from nf_mlp.hip)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c3; mkdir -p $O
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in "" _x_nont _x_blocked _x_l2sink _x_nostore _r03 ""; do echo "== lib${v:-_new}"; NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 300 python tools/time_train_split.py bf16x3 f16x3 2>&1 | grep "paper"; done > $O/store_ablation.txt; cat $O/store_ablation.txt
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/pytest_gpu.txt; tail -6 $O/pytest_gpu.txt
