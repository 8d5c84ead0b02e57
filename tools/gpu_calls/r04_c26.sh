# round 4, call 26: timeline of one f16x3 and one bf16x3 training iteration (kernel trace): the non-MLP part
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c26; mkdir -p $O
for PREC in f16x3 bf16x3; do
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/bench.py --mode train --precision $PREC --steps 6 --warmup 3 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/$O/bench_train_$PREC.json 2> $GRAFT_REPO_ROOT/$O/bench_train_$PREC.err
cd $GRAFT_REPO_ROOT
python - <<'PY' > $O/timeline_$PREC.txt
import sqlite3, glob
db = glob.glob('/tmp/kt/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, start, end, grid_x from kernels order by start").fetchall()
# find the last two launches of the dW kernel's last call per iteration: use k_adam / optimizer kernel as the iteration boundary
names = [r[0] for r in rows]
idx = [i for i, n in enumerate(names) if 'adam' in n.lower()]
print("kernels", len(rows), "adam launches", len(idx))
a, b = idx[-3], idx[-2]
t0 = rows[a][2]
busy = 0
for i in range(a + 1, b + 1):
    n, s, e, g = rows[i]
    gap = (s - rows[i - 1][2]) / 1e3
    busy += (e - s)
    print(f"{(s - t0) / 1e3:9.1f} us  +{gap:6.1f} gap  {(e - s) / 1e3:8.1f} us  grid {g:8d}  {n.split('(')[0][:90]}")
print("iteration span %.1f us, kernel-busy %.1f us" % ((rows[b][2] - t0) / 1e3, busy / 1e3))
PY
tail -c 600 $O/bench_train_$PREC.json; echo; cat $O/timeline_$PREC.txt | tail -75

done
