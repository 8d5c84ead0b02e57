# round 5, call 1: the driver's EXACT bench command on a fresh box, bracketed by the stand-alone store-pattern probe and the per-kernel
# training times (is the slow-store condition a property of the box, or of what ran before?), plus the new tests of this round
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c1; mkdir -p $O
smi() { (rocm-smi --showmemorypartition --showcomputepartition --showpower --showmaxpower --showtemp --showclocks --showperflevel --showmeminfo vram 2>&1 | grep -v "^$" | head -60) > $1; }
smi $O/smi_before.txt
tools/micro/store_bw > $O/store_bw_before.json 2>&1; cat $O/store_bw_before.json
timeout 300 python tools/time_train_split.py bf16x3 f16x3 f32 > $O/train_before.txt 2>&1; tail -4 $O/train_before.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc $?"; tail -n 1 $O/bench.out | cut -c1-6000
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
tools/micro/store_bw > $O/store_bw_after.json 2>&1; cat $O/store_bw_after.json
timeout 300 python tools/time_train_split.py bf16x3 f16x3 f32 > $O/train_after.txt 2>&1; tail -4 $O/train_after.txt
smi $O/smi_after.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_launchers.py -m gpu -q --tb=short -k "resample or bare_gpus" 2>&1 | grep -v Warning | tail -15 > $O/pytest_new.txt; tail -5 $O/pytest_new.txt
tail -5 $O/bench.err
