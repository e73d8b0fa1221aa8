#!/usr/bin/env python3
"""GPU idle time in a rocprofv3 kernel trace (rocpd sqlite): the kernels (and memory copies, when traced) of the LAST `--window-ms`
of the run are merged into busy intervals; prints busy / idle totals and the largest gaps with the kernel in front of and behind
each — the places where the host makes the GPU wait (a d2h of a count, an allocation, a launch chain).
Usage: tools/rocpd_timeline.py results.db [--window-ms 30] [--gaps 25] [--min-gap-us 5]"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--window-ms", type=float, default=0.0, help="only the last so many ms of the trace (0 = all)")
    ap.add_argument("--gaps", type=int, default=25)
    ap.add_argument("--min-gap-us", type=float, default=5.0)
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    ev = [(s, e, n.replace("(anonymous namespace)::", "").split("(")[0][-60:]) for s, e, n in db.execute(f"select start, end, {namecol} from kernels")]
    try:
        mc = [r[1] for r in db.execute("pragma table_info(memory_copies)")]
        if mc:
            ev += [(s, e, "copy:" + str(n)) for s, e, n in db.execute("select start, end, name from memory_copies")]
    except sqlite3.Error:
        pass
    ev.sort()
    if not ev:
        print("no events")
        return
    t_end = max(e for _, e, _ in ev)
    if a.window_ms:
        ev = [x for x in ev if x[0] >= t_end - a.window_ms * 1e6]
    t0 = ev[0][0]
    busy = 0
    gaps = []
    cur_s, cur_e, last_name = ev[0]
    for s, e, n in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e - t0, last_name, n))
            cur_s, cur_e, last_name = s, e, n
        else:
            if e > cur_e:
                cur_e, last_name = e, n
    busy += cur_e - cur_s
    span = cur_e - t0
    print(f"span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms in {len(gaps)} gaps ({len(ev)} events)")
    big = [g for g in gaps if g[0] >= a.min_gap_us * 1e3]
    print(f"gaps >= {a.min_gap_us} us: {len(big)}, {sum(g[0] for g in big)/1e6:.3f} ms")
    hist = {}
    for g, at, before, after in gaps:
        k = (before, after)
        c = hist.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += g
    print("\n| idle us (sum) | count | after kernel | before kernel |\n|---|---|---|---|")
    for (b, af), (c, s) in sorted(hist.items(), key=lambda kv: -kv[1][1])[: a.gaps]:
        print(f"| {s/1e3:.1f} | {c} | {b} | {af} |")


if __name__ == "__main__":
    main()
