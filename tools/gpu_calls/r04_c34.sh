# round 4, call 34: fc_rgb with all eight weight fragments and B fragments in flight: fingerprints and fine-launch time against the previous library
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c34; mkdir -p $O
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in _prev "" _prev "" _prev ""; do echo "== lib${v:-_new}"; NERFACE_HIP_LIB=$L/libnerface_hip$v.so TIME_MLP_HASH=1 TIME_MLP_ONLY_F32=1 timeout 300 python tools/time_mlp.py 2>&1 | grep "hash\|f32 \|Error" | cut -c1-120; done > $O/fc_rgb_ab.txt; cat $O/fc_rgb_ab.txt
