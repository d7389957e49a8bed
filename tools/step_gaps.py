#!/usr/bin/env python3
"""Where a launch-heavy step's wall time goes (rocprofv3 --kernel-trace, rocpd SQLite): over a steady-state window of N periods of the
marker kernel, the busy time (union of kernel intervals), the idle time, kernels per period, and the idle gaps grouped by the kernel
that FOLLOWS them (the launch that was late).

    python tools/step_gaps.py <k_results.db> --marker k_c1_imgstats --per-step 6 [--steps 10] [--dump 0]
"""
import argparse
import collections
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--marker", required=True)
    ap.add_argument("--per-step", type=int, default=1, help="launches of the marker per step")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--dump", type=int, default=0, help="print this many kernels of the window")
    a = ap.parse_args()
    c = sqlite3.connect(a.db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if a.marker in r[0]]
    n = len(marks) // a.per_step
    first = marks[(n // 2) * a.per_step]
    last = marks[min(len(marks) - 1, (n // 2 + a.steps) * a.per_step)]
    win = rows[first:last]
    t0, t1 = win[0][1], rows[last][1]
    busy, cur_s, cur_e = 0, win[0][1], win[0][2]
    gaps = collections.defaultdict(lambda: [0, 0.0])
    durs = collections.defaultdict(lambda: [0, 0.0])
    for nm, s, e in win[1:] + [rows[last]]:
        if s > cur_e:
            busy += cur_e - cur_s
            k = nm.split("(")[0].replace("void ", "")[:60]
            gaps[k][0] += 1; gaps[k][1] += (s - cur_e) / 1e3
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    for nm, s, e in win:
        k = nm.split("(")[0].replace("void ", "")[:60]
        durs[k][0] += 1; durs[k][1] += (e - s) / 1e3
    wall = (t1 - t0) / 1e3
    print("# window: %d steps, %.1f us per step wall, busy %.1f us per step (%.1f %%), %d kernels per step"
          % (a.steps, wall / a.steps, busy / 1e3 / a.steps, 100.0 * busy / 1e3 / wall, len(win) / a.steps))
    print("# idle gaps by the kernel that follows them (per step): count, total us, avg us")
    for k, (cnt, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
        print("  %6.1f %9.1f %7.1f  %s" % (cnt / a.steps, tot / a.steps, tot / cnt, k))
    print("# kernel time per step: count, total us, avg us")
    for k, (cnt, tot) in sorted(durs.items(), key=lambda kv: -kv[1][1])[:45]:
        print("  %6.1f %9.1f %7.1f  %s" % (cnt / a.steps, tot / a.steps, tot / cnt, k))
    for nm, s, e in win[:a.dump]:
        print("  %9.1f %9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, nm.split("(")[0][:80]))


if __name__ == "__main__":
    main()
