# GPU call 12 (round 3): one-launch Adam: its test, the launcher round trip, lcode tests, training timing (both families, three arithmetics of the paper model)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_lcode.py "tests/test_gpu_launchers.py::test_train_then_eval_roundtrip" "tests/test_gpu_launchers.py::test_launchers_second_model_family" -x -q -m gpu 2>&1 | tail -4
for args in "--family paper" "--family paper" "--family lcode" "--family paper --precision f16x3"; do timeout 200 python bench.py --mode train $args --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$args ms/iter', d['ms_per_step'], 'mlp', d['roofline'].get('ms_both_launches'))"; done
