# GPU call 7 (round 3): nt stores + early staging read in dW: timing, A/B, backward tests; kernel trace of the f32 training iteration
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c7
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
for v in "" _nostage3 ""; do
  echo "== variant '$v'"
  NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 120 python tools/time_train_f32.py 10 2>&1 | grep -v amdgpu.ids
done > gpurun_out/c7/variants.log 2>&1
cat gpurun_out/c7/variants.log
timeout 200 python tools/ab_train_f32.py --odd 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tail -4
timeout 600 python -m pytest tests/test_gpu_backward.py -x -q -m gpu 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 40 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/c7/train_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/c7/prof.log; echo "prof rc=$?"
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $DB > $GRAFT_REPO_ROOT/gpurun_out/c7/train_iter_kernel_stats.md 2>&1
head -60 $GRAFT_REPO_ROOT/gpurun_out/c7/train_iter_kernel_stats.md | cut -c1-130
python -c "
import json
d=json.loads([l for l in open('$GRAFT_REPO_ROOT/gpurun_out/c7/train_line.json') if l.startswith('{')][-1]); print('train ms/iter under profiler', d['ms_per_step'])"
