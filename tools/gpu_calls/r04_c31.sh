# round 4, call 31: k_paper_grad_unpack with the d-latent reduction on all 256 threads of its workgroup: backward tests, kernel duration in a traced run
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c31; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_bf16.py tests/test_gpu_f16.py -q -m gpu --tb=short 2>&1 | grep -v Warning | tail -6 > $O/pytest.txt; tail -3 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --mode train --precision f32 --steps 6 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee $O/unpack.txt
import sqlite3, glob
db = glob.glob('/tmp/kt/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
for pat in ('%k_paper_grad_unpack%', '%k_grad_reduce%'):
    d = [r[0] / 1e3 for r in c.execute("select duration from kernels where name like ?", (pat,)).fetchall()]
    print(pat, len(d), "launches, avg %.1f us, min %.1f, max %.1f" % (sum(d) / len(d), min(d), max(d)))
PY
