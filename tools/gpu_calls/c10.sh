# GPU call 10 (round 3): second model family on the new exact-f32 training kernels: its tests, the launcher test of that family, train timing of both families
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c10
timeout 900 python -m pytest tests/test_gpu_lcode.py "tests/test_gpu_launchers.py::test_launchers_second_model_family" tests/test_gpu_backward.py -x -q -m gpu --durations=5 2>&1 | tail -12
for fam in lcode paper; do timeout 200 python bench.py --mode train --family $fam --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$fam train f32 ms/iter', d['ms_per_step'], 'mlp', d['roofline'].get('ms_both_launches'))"; done
