# round 5, call 4: the new end-to-end gradient test (full tensors vs reference autograd, no-flip case) in all three arithmetics; what the
# amdgpu sysfs nodes offer on the box; a default bench run carrying the new power / clock probe.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c4; mkdir -p $O
(for d in /sys/class/drm/card*/device; do echo "== $d"; ls $d | tr '\n' ' '; echo; ls $d/hwmon/*/ | tr '\n' ' '; echo;
  for f in power_dpm_force_performance_level pp_dpm_sclk pp_dpm_fclk pp_dpm_mclk pp_dpm_socclk current_compute_partition current_memory_partition pp_power_profile_mode; do echo "-- $f"; cat $d/$f 2>&1 | head -12; done;
  for f in $d/hwmon/*/*; do [ -f $f ] && echo "$f = $(cat $f 2>&1 | head -c 80)"; done; done) > $O/sysfs.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_backward.py -m gpu -q -s -k "full_tensors" 2>&1 | grep -v Warning | tail -12 > $O/pytest_noflip.txt; cat $O/pytest_noflip.txt
timeout 900 python3 bench.py > $O/bench.out 2> $O/bench.err; echo "bench rc $?"; tail -n 1 $O/bench.out | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(len(json.dumps(d))); print({k:v for k,v in d['summary'].items() if 'power' in k or 'sclk' in k or 'fclk' in k or 'perf' in k or 'train_ms' in k or 'split' in k})"
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
python3 -c "import json; d=json.load(open('$O/bench_detail.json'))['bench_detail']; print(json.dumps(d['config']['device'].get('power'), indent=1)[:3000])"
tail -5 $O/bench.err
