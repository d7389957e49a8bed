#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace run (rocpd SQLite): for the steady state of the fused pipeline print every kernel
with start / end relative to the window, the queue it ran on, and the idle gaps of the projection kernels' queue.

    python tools/timeline.py <k_results.db> [--match k_project] [--skip 0.5] [--rows 80]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--match", default="k_project")
    ap.add_argument("--skip", type=float, default=0.6, help="start the window at this fraction of the matching launches")
    ap.add_argument("--rows", type=int, default=70)
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print("# columns:", ", ".join(cols))
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    scol = "stream_id" if "stream_id" in cols else qcol
    sel = "select name, start, end, %s, %s from kernels order by start" % (qcol or "0", scol or "0")
    rows = c.execute(sel).fetchall()
    big = [r for r in rows if a.match in r[0] and (r[2] - r[1]) > 100000]
    if not big:
        print("no matching kernels"); return
    t0 = big[int(len(big) * a.skip)][1]
    win = [r for r in rows if r[1] >= t0][:a.rows]
    print("# %8s %8s %8s  %-6s %-6s %s" % ("start_us", "end_us", "dur_us", "queue", "stream", "kernel"))
    for n, s, e, q, st in win:
        nm = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        print("  %8.1f %8.1f %8.1f  %-6s %-6s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, st, nm))
    # gaps between consecutive matching launches over the whole steady state
    gaps, durs = [], []
    for x, y in zip(big[2:-2], big[3:-1]):
        gaps.append((y[1] - x[2]) / 1e3); durs.append((x[2] - x[1]) / 1e3)
    if gaps:
        gaps_s = sorted(gaps)
        print("# %d launches of %s: avg duration %.1f us, avg gap to the next launch %.1f us (median %.1f, p90 %.1f, max %.1f); period %.1f us"
              % (len(durs), a.match, sum(durs) / len(durs), sum(gaps) / len(gaps), gaps_s[len(gaps) // 2], gaps_s[int(len(gaps) * 0.9)],
                 gaps_s[-1], (big[-2][1] - big[2][1]) / 1e3 / (len(big) - 4)))


if __name__ == "__main__":
    main()
