"""Where does a frame's wall time go besides the kernels?  Reads a rocprofv3 --kernel-trace CSV and prints, for the longest
stretch of back-to-back dispatches, the GPU-busy time, the idle time between kernels and the largest gaps with their neighbours.
    cd /tmp && rocprofv3 --kernel-trace -f csv -d out -o t -- python bench.py --no-extras --no-cpu-baseline --precision f16x3
    python tools/trace_gaps.py out/**/t_kernel_trace.csv [min_gap_us]"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 20e3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]) for r in rows))
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = 0
cur_s, cur_e = ev[0][0], ev[0][1]
gaps = []
prev = ev[0]
for s, e, n in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, prev[2], n, cur_e - t0))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev = (s, e, n)
busy += cur_e - cur_s
print(f"span {1e-6 * (t1 - t0):.1f} ms, GPU busy {1e-6 * busy:.1f} ms, idle {1e-6 * (t1 - t0 - busy):.1f} ms in {len(gaps)} gaps")
hist = {}
for g, a, b, at in gaps:
    if g >= min_gap:
        k = (a, b)
        hist.setdefault(k, [0, 0])
        hist[k][0] += 1
        hist[k][1] += g
for (a, b), (cnt, tot) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{cnt:5d} gaps, {1e-6 * tot:8.2f} ms total, {1e-3 * tot / cnt:8.1f} us each: {a}  ->  {b}")
small = sum(g for g, *_ in gaps if g < min_gap)
print(f"gaps below {min_gap / 1e3:.0f} us: {1e-6 * small:.2f} ms")
