# round 4, call 29: PMC counters of k_resample_merge_small (instruction mix per wave): is it VALU-issue-bound as the ISA count says?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04_c29; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_ANY"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc$i -o p -- python $GRAFT_REPO_ROOT/tools/resample_check.py > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee $O/resample_pmc.txt
import sqlite3, glob
from collections import defaultdict
for i in range(1, 6):
    dbs = glob.glob(f'/tmp/pmc{i}/**/*.db', recursive=True)
    if not dbs: print("pass", i, "no db"); continue
    c = sqlite3.connect(dbs[0])
    rows = c.execute("select kernel_name, grid_size_x, counter_name, value from counters_collection where kernel_name like '%resample_merge_small%'").fetchall()
    agg = defaultdict(list)
    for n, g, cn, v in rows: agg[(g, cn)].append(v)
    for (g, cn), v in sorted(agg.items()):
        if g == 4194304: print(f"grid {g} {cn}: launches {len(v)} mean {sum(v)/len(v):.6g}")
PY
