# round 4, call 28: k_stream_absmax on 256 workgroups with four entries in flight: f16 parity tests, its duration in a traced f16x3 training run
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c28; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_f16.py tests/test_gpu_lcode.py -q -m gpu --tb=short 2>&1 | grep -v Warning | tail -8 > $O/pytest.txt; tail -4 $O/pytest.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --mode train --precision f16x3 --steps 6 --warmup 3 --no-cpu-baseline --no-extras > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee $O/absmax.txt
import sqlite3, glob
db = glob.glob('/tmp/kt/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
for pat in ('%k_stream_absmax%', '%k_pack_split_f16%'):
    rows = c.execute("select duration from kernels where name like ?", (pat,)).fetchall()
    d = [r[0] / 1e3 for r in rows]
    print(pat, len(d), "launches, avg %.1f us, min %.1f, max %.1f" % (sum(d) / len(d), min(d), max(d)))
PY
