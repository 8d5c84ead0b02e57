# GPU call 9 (round 3): ILP'd choice kernels, reduction without slab zero-fill, wider unpack grid: kernel + backward tests, train timing, short trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c9
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_backward.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do timeout 200 python bench.py --mode train --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('train f32 ms/iter', d['ms_per_step'], 'mlp', d['roofline']['ms_both_launches'])"; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 40 --warmup 5 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/c9/prof.log; echo "prof rc=$?"
DB=$(find /tmp/prof_tr -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $DB > $GRAFT_REPO_ROOT/gpurun_out/c9/train_iter_kernel_stats.md 2>&1
grep "k_choice\|at::native:: |\|fillBuffer\|grad_unpack\|grad_reduce" $GRAFT_REPO_ROOT/gpurun_out/c9/train_iter_kernel_stats.md | cut -c1-110
