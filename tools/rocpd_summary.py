#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database: per-kernel call count / total / average duration (the same table
`--stats` prints) and, when the run collected PMC counters, the per-kernel counter sums per dispatch.
Usage: tools/rocpd_summary.py results.db [--md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {namecol} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for n, c, s, a, mn, mx in rows:
        n = n.replace("(anonymous namespace)::", "").split("(")[0][-90:]
        print(f"| {n} | {c} | {s/1e6:.3f} | {a/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {100*s/tot:.1f} |")
    try:
        pc = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        if pc:
            kn = "kernel_name" if "kernel_name" in pc else namecol
            cn = "counter_name" if "counter_name" in pc else "name"
            q = f"select {kn}, {cn}, count(*), sum(value) from counters_collection group by {kn}, {cn} order by 1, 2"
            res = db.execute(q).fetchall()
            if res:
                print("\n| kernel | counter | dispatches | sum | per dispatch |")
                print("|---|---|---|---|---|")
                for k, c, n, s in res:
                    print(f"| {k.replace('(anonymous namespace)::', '').split('(')[0][-70:]} | {c} | {n} | {s:.6g} | {s/n:.6g} |")
    except sqlite3.Error as e:
        print("no counters:", e)


if __name__ == "__main__":
    main()
