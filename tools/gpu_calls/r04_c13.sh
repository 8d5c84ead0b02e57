# round 4, call 13: the layer-streamed exact-f32 dX chain against the round-3 chain: dZ / gradient fingerprints and stage times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c13; mkdir -p $O
for old in 1 0 1 0; do echo "== NERFACE_F32_CHAIN_OLD=$old"; NERFACE_F32_CHAIN_OLD=$old timeout 300 python tools/chain_check.py 2>&1 | grep "sha1\|f32 @"; done > $O/chain_ab.txt; cat $O/chain_ab.txt
