# round 5, call 14: the multi-rank bench path WITH its extras (the other arithmetics incl. f16x2, the training leg with the gradient
# all-reduce, the rank-0 legs) -- 2 ranks on one GPU over gloo, launched by bench.py itself (--gpus 2)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c14; mkdir -p $O
( time NERFACE_DIST_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 2 --warmup 1 --train-steps 5 --no-cpu-baseline > $O/bench2.out 2> $O/bench2.err ) 2> $O/time.txt; echo "rc $?"; grep real $O/time.txt
tail -n 1 $O/bench2.out | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['summary']; print(d['n_gpus'], d['ranks_seen'], d['value'], {k: s.get(k) for k in ('split_f16x2_rays_s','split_f16_rays_s','train_ms_per_iter_f32','train_ms_per_iter_bf16x3','train_allreduce_us','train_bytes_allreduced','train_ranks_seen')})"
tail -3 $O/bench2.err
