# round 5, call 23: default bench run with the four-frame gate check on the line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c23; mkdir -p $O
timeout 600 python3 bench.py > $O/bench.out 2> $O/bench.err; echo "rc $?"
tail -n 1 $O/bench.out | python3 -c "import sys,json; t=sys.stdin.read(); d=json.loads(t); s=d['summary']; print(len(t), d['value'], {k: v for k, v in s.items() if 'gate' in k or 'frame_' in k})"
tail -2 $O/bench.err
