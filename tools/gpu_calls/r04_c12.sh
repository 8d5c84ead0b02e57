# round 4, call 12: A-fragment tile group (NFB_TILE_GROUP) sweep of the split training kernels
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c12; mkdir -p $O
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in "" _tg2 _tg4 _tg8 "" _tg2 _tg4 _tg8; do echo "== lib${v:-_default}"; NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 300 python tools/time_train_split.py bf16x3 f16x3 2>&1 | grep "paper\|box"; done > $O/tg_sweep.txt; cat $O/tg_sweep.txt
