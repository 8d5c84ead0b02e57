# round 4, call 15: nf_tail_dz schedule variants (lag of the masked slab write behind the MFMAs; VALU group hint)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c15; mkdir -p $O
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in "" _lag1 _lag3 _lag4 _valu0 "" _lag1 _lag3 _lag4 _valu0; do echo "== lib${v:-_default}"; NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 300 python tools/chain_check.py 2>&1 | grep "2048x128\|f32 @"; done > $O/tail_dz.txt; cat $O/tail_dz.txt
