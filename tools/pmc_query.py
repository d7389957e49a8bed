#!/usr/bin/env python3
"""Print per-kernel averages of every counter in a rocprofv3 --pmc run (rocpd SQLite)."""
import collections, sqlite3, statistics, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "%"
rows = c.execute("select kernel_name, counter_name, value, duration from counters_collection where kernel_name like ?", (pat,)).fetchall()
agg = collections.defaultdict(list)
for k, cn, v, d in rows:
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
    agg[(k, cn)].append((v, d))
for (k, cn), vals in sorted(agg.items()):
    big = [x for x in vals if x[1] > 20000] or vals
    print("%-52s %-28s n=%-4d avg=%-16.1f dur_us=%.1f" % (k, cn, len(big), statistics.mean(v for v, _ in big), statistics.mean(d for _, d in big) / 1e3))
