#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2 rocpd SQLite) outputs into the text files committed under profiles/.

  python tools/rocpd_summary.py stats  <results.db>              -> per-kernel time table (+ per grid size)
  python tools/rocpd_summary.py pmc    <results.db> [<results.db> ...]  -> per-kernel counter sums / per-launch means
"""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0].replace("void ", "")


def stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, grid_x, workgroup_x, duration, vgpr_count, accum_vgpr_count, sgpr_count, lds_size from kernels").fetchall()
    tot = sum(r[3] for r in rows)
    agg = defaultdict(list)
    meta = {}
    for name, gx, wx, dur, v, a, s, lds in rows:
        agg[(short(name), gx)].append(dur)
        meta[short(name)] = (wx, v, a, s, lds)
    byk = defaultdict(list)
    for (n, gx), d in agg.items():
        byk[n] += d
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
    print(f"total kernel time {tot/1e6:.3f} ms over {len(rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % | wg | vgpr | agpr | sgpr | lds |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for n, d in sorted(byk.items(), key=lambda kv: -sum(kv[1])):
        wx, v, a, s, lds = meta[n]
        print(f"| {n} | {len(d)} | {sum(d)/1e6:.3f} | {sum(d)/len(d)/1e3:.1f} | {min(d)/1e3:.1f} | {max(d)/1e3:.1f} | {100*sum(d)/tot:.2f} | {wx} | {v} | {a} | {s} | {lds} |")
    print("\n## per (kernel, grid size) for kernels above 1 % of the time\n")
    print("| kernel | grid_x (threads) | calls | avg us | min us | max us |")
    print("|---|---|---|---|---|---|")
    for (n, gx), d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if sum(byk[n]) / tot > 0.01:
            print(f"| {n} | {gx} | {len(d)} | {sum(d)/len(d)/1e3:.1f} | {min(d)/1e3:.1f} | {max(d)/1e3:.1f} |")
    # A persistent-grid kernel launches the same grid for every problem size: its launches are told apart by their durations.
    print("\n## duration clusters (a new cluster where the sorted durations jump by more than 25 %), kernels above 1 % of the time\n")
    print("| kernel | grid_x (threads) | cluster | calls | avg us | min us | max us |")
    print("|---|---|---|---|---|---|---|")
    for (n, gx), d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if sum(byk[n]) / tot <= 0.01:
            continue
        d = sorted(d)
        clusters, cur_c = [], [d[0]]
        for x in d[1:]:
            if x > 1.25 * cur_c[-1]:
                clusters.append(cur_c)
                cur_c = [x]
            else:
                cur_c.append(x)
        clusters.append(cur_c)
        if len(clusters) > 1:
            for k, c in enumerate(clusters):
                print(f"| {n} | {gx} | {k} | {len(c)} | {sum(c)/len(c)/1e3:.1f} | {min(c)/1e3:.1f} | {max(c)/1e3:.1f} |")


def pmc(dbs):
    print("# rocprofv3 --pmc summary (separate passes per counter group)\n")
    print("| db | kernel | grid_x | counter | launches | mean per launch | sum |")
    print("|---|---|---|---|---|---|---|")
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select kernel_name, grid_size_x, counter_name, value from counters_collection").fetchall()
        agg = defaultdict(list)
        for n, gx, c, v in rows:
            agg[(short(n), gx, c)].append(v)
        for (n, gx, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:24]:
            print(f"| {db} | {n} | {gx} | {c} | {len(v)} | {sum(v)/len(v):.4g} | {sum(v):.6g} |")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
