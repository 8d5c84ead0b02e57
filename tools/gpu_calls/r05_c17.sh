# round 5, call 17: second family in the four arithmetics (launch + whole frame), zero-operand probe incl. f16x2, bench with f16x2 as the
# headline arithmetic (sanity of that path), a default bench run (PMC traffic of the f16x2 kernel now on the line)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c17; mkdir -p $O
timeout 300 python tools/time_lcode.py 2>&1 | grep "^lcode" | tee $O/time_lcode.txt
timeout 300 python tools/zero_data_probe.py 2>&1 | grep "operands" | tail -8 | tee $O/zero_data.txt
timeout 600 python3 bench.py --precision f16x2 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_x2_headline.out 2> $O/bench_x2_headline.err; echo "rc $?"; tail -n 1 $O/bench_x2_headline.out | cut -c1-700
timeout 900 python3 bench.py > $O/bench.out 2> $O/bench.err; echo "rc $?"; tail -n 1 $O/bench.out | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(len(json.dumps(d)), d['value'], d['roofline']['frac'])"
python3 -c "import json; d=json.load(open('gpurun_out/bench_detail.json'))['bench_detail']; r=d['split_f16x2']['roofline']; print('f16x2 traffic', r.get('traffic'), r.get('algorithmic_hbm_bytes_per_launch'), r.get('avg_launch_ms'))"
