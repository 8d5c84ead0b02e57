# round 5, call 2: is the slow-store condition of the driver's boxes what the GPU test suite leaves behind?  The driver runs pytest -m gpu
# (51 python processes) and smoke() before bench.py; every builder call so far ran bench.py on a virgin box.  Here: probe, the full suite,
# smoke, probe, then the driver's exact bench command, probe.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_c2; mkdir -p $O
smi() { (rocm-smi --showmemorypartition --showcomputepartition --showpower --showmaxpower --showtemp --showclocks --showperflevel --showmeminfo vram 2>&1 | grep -v "^$" | head -60) > $1; }
smi $O/smi_before.txt
env | grep -E '^(HSA|HIP|ROC|GPU|PYTORCH|AMD|NCCL|RCCL)' > $O/env.txt
tools/micro/store_bw > $O/store_bw_0_virgin.json 2>&1; cat $O/store_bw_0_virgin.json
timeout 300 python tools/time_train_split.py bf16x3 f16x3 f32 > $O/train_0_virgin.txt 2>&1; tail -4 $O/train_0_virgin.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warning | tail -15 > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
tools/micro/store_bw > $O/store_bw_1_after_suite.json 2>&1; cat $O/store_bw_1_after_suite.json
timeout 300 python tools/time_train_split.py bf16x3 f16x3 f32 > $O/train_1_after_suite.txt 2>&1; tail -4 $O/train_1_after_suite.txt
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc $?"; tail -n 1 $O/bench.out | cut -c1-6000
cp gpurun_out/bench_detail.json $O/bench_detail.json 2>/dev/null
tools/micro/store_bw > $O/store_bw_2_after_bench.json 2>&1; cat $O/store_bw_2_after_bench.json
smi $O/smi_after.txt
tail -5 $O/bench.err
