"""Engine clock each kernel of a command actually held: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs by rocprofv3) / dispatch
duration, from one rocprofv3 PMC pass (bench.pmc_pass_rows).  Usage: python tools/kernel_clocks.py <tools-script> [args...]"""
import os, shutil, sys, tempfile
from collections import defaultdict
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
prof, why = bench._pmc_guard()
assert prof, why
tmp = tempfile.mkdtemp(prefix="nf_clk_")
rows, err = bench.pmc_pass_rows(prof, tmp, "GRBM_GUI_ACTIVE", sys.argv[1], sys.argv[2:], 300)
assert rows is not None, err
agg = defaultdict(list)
for name, gx, v, d in rows:
    if d:
        agg[(name.split("(")[0].replace("void ", ""), gx)].append((v / (d * 1e-9) / 8 / 1e6, d / 1e3))
print("| kernel | grid | dispatches | clock MHz (per dispatch) | duration us (under PMC) |")
print("|---|---|---|---|---|")
for (n, gx), xs in sorted(agg.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    if sum(d for _, d in xs) > 200:
        print(f"| {n} | {gx} | {len(xs)} | {' '.join(f'{c:.0f}' for c, _ in xs[:6])} | {' '.join(f'{d:.0f}' for _, d in xs[:6])} |")
shutil.rmtree(tmp, ignore_errors=True)
