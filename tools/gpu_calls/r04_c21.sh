# round 4, call 21: k_resample_merge_small: sign-domain sort (one v_min per exchange), bin mid-points from the rank test's depths; rocprof kernel time
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c21; mkdir -p $O
L=$GRAFT_REPO_ROOT/4d-facial-avatars_amd/lib
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -k "resample or sample_pdf or sort" 2>&1 | grep -v Warning | tail -25 > $O/pytest.txt; cat $O/pytest.txt | tail -25
for v in _prev "" _prev ""; do echo "== lib${v:-_new}"; NERFACE_HIP_LIB=$L/libnerface_hip$v.so timeout 300 python tools/resample_check.py 2>&1 | grep "resample\|Error" | head -8; done > $O/resample_ab.txt; cat $O/resample_ab.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $GRAFT_REPO_ROOT/tools/resample_check.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r04_c21/kernel_trace.txt
import sqlite3, glob
db = glob.glob('/tmp/kt/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, grid_x, duration from kernels where name like '%resample_merge_small%' order by start").fetchall()
by = {}
for n, g, d in rows: by.setdefault(g, []).append(d / 1e3)
for g, ds in by.items(): print("k_resample_merge_small grid_x", g, "launches", len(ds), "avg %.1f us min %.1f max %.1f" % (sum(ds)/len(ds), min(ds), max(ds)))
PY
