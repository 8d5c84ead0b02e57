for v in _old "" _old ""; do
  echo "== lib$v"
  NERFACE_HIP_LIB=$PWD/4d-facial-avatars_amd/lib/libnerface_hip$v.so python bench.py --mode train --steps 80 --warmup 10 --precision bf16x3 --family lcode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lcode bf16x3 train ms/iter', round(d['ms_per_step'],3), 'mlp-only', round(d['roofline']['ms_both_launches'],3))"
done
