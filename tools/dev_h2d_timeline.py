"""Dev probe: kernel timeline of ONE PCIe-inclusive insert (pinned host bases -> cblx_insert_seqs + flush) out of a rocprofv3 results
database: span, busy time, the gaps between consecutive kernels. Usage: rocprofv3 --kernel-trace -d D -o h -- python3 tools/dev_h2d_timeline.py run;
then python3 tools/dev_h2d_timeline.py report D/h_results.db"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import numpy as np, torch, cbl_amd
    from cbl_amd import synth
    NR, L = 10_000_000, 150
    d_b, d_o = synth.reads_torch(42, NR, L, device="cuda")
    hb = torch.empty(NR * L, dtype=torch.uint8, pin_memory=True); ho = torch.empty(NR + 1, dtype=torch.int64, pin_memory=True)
    hb.copy_(d_b[: NR * L]); ho.copy_(d_o); torch.cuda.synchronize()
    g = cbl_amd.CBL(31, 24)
    nb_, no_ = hb.numpy(), ho.numpy().view(np.uint64)
    ts = []
    for rep in range(int(os.environ.get("REPS", 3))):
        g.clear(); torch.cuda.synchronize()
        time.sleep(0.5)  # a visible gap in the trace in front of every repetition
        t0 = time.perf_counter(); g.insert_seqs(nb_, no_); t1 = time.perf_counter(); g.flush(); t2 = time.perf_counter()
        ts.append(t2 - t0)
        print("insert_seqs %.2f + flush %.2f = %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3), flush=True)
    print("best %.2f median %.2f ms" % (min(ts[1:]) * 1e3, sorted(ts[1:])[len(ts[1:]) // 2] * 1e3))
else:
    import sqlite3
    db = sqlite3.connect(sys.argv[2])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = db.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    # the last repetition = kernels after the last gap of more than 300 ms
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][1] - rows[i - 1][2] > 300e6: cut = i
    rows = rows[cut:]
    t0 = rows[0][1]
    span = (max(r[2] for r in rows) - t0) / 1e6
    busy = 0.0; cur_end = t0; gaps = []
    for n, s, e in rows:
        if s > cur_end:
            gaps.append(((s - cur_end) / 1e3, (cur_end - t0) / 1e6, n.split("(")[0][-60:]))
        busy += max(0, e - max(s, cur_end)) / 1e6
        cur_end = max(cur_end, e)
    print("kernels %d, span %.2f ms, busy (union) %.2f ms, idle %.2f ms" % (len(rows), span, busy, span - busy))
    gaps.sort(reverse=True)
    for g_us, at_ms, nxt in gaps[:25]: print("  gap %8.1f us at %6.2f ms before %s" % (g_us, at_ms, nxt))
    # time line of the big kernels
    for n, s, e in rows:
        if e - s > 300e3: print("  %6.2f .. %6.2f ms  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, n.split("(")[0][-70:]))
