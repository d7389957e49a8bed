#!/usr/bin/env python3
"""Condense rocprofv3 output (rocpd SQLite: `rocprofv3 --kernel-trace --stats` and `--pmc ...` runs) into
the small text summaries committed under profiles/.

    python tools/prof_summary.py stats <bench_results.db>          > profiles/rNN_kernel_stats.txt
    python tools/prof_summary.py pmc   <fetch.db> <write.db>       > profiles/rNN_pmc.txt  (also writes pmc_latest.json)
"""
import collections
import json
import os
import sqlite3
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:90]


def stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count from kernels").fetchall()
    agg = collections.defaultdict(list)
    meta = {}
    for name, dur, gx, wx, lds, vg, sg in rows:
        agg[short(name)].append(dur)
        meta[short(name)] = (wx, lds, vg, sg)
    tot = sum(sum(v) for v in agg.values())
    print("# rocprofv3 --kernel-trace --stats summary (durations in us)")
    print("%-92s %7s %12s %10s %10s %10s %6s %6s %7s %5s %5s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "wg", "lds", "vgpr", "sgpr"))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        wx, lds, vg, sg = meta[k]
        print("%-92s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %6d %7d %5d %5d" % (
            k, len(v), sum(v) / 1e3, statistics.mean(v) / 1e3, min(v) / 1e3, max(v) / 1e3, 100.0 * sum(v) / tot, wx, lds, vg, sg))


def pmc(fetch_db, write_db):
    out = {}
    for label, db in (("FETCH_SIZE", fetch_db), ("WRITE_SIZE", write_db)):
        c = sqlite3.connect(db)
        rows = c.execute("select kernel_name, value, duration, grid_size from counters_collection where counter_name=?", (label,)).fetchall()
        agg = collections.defaultdict(list)
        for k, v, d, g in rows:
            agg[(short(k), g)].append((v, d))
        out[label] = agg
    print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KB per dispatch as reported")
    print("# gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports 1/2 of the bytes of a wide coalesced")
    print("# streaming read -> HBM read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is 1:1 here (k_synth writes exactly B*4XYZ).")
    print("%-70s %10s %6s %16s %16s %12s %16s" % ("kernel", "grid", "n", "FETCH_KB(avg)", "WRITE_KB(avg)", "avg_us", "HBM_bytes(corr)"))
    keys = sorted(set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"]), key=lambda k: -sum(d for _, d in out["FETCH_SIZE"].get(k, [(0, 0)])))
    summary = {}
    for k in keys:
        f = [x for x in out["FETCH_SIZE"].get(k, []) if x[1] > 20000][1:] or [x for x in out["FETCH_SIZE"].get(k, []) if x[1] > 20000]
        w = [x for x in out["WRITE_SIZE"].get(k, []) if x[1] > 20000][1:] or [x for x in out["WRITE_SIZE"].get(k, []) if x[1] > 20000]
        if not f and not w:
            continue
        fk = statistics.mean(v for v, _ in f) if f else 0.0
        wk = statistics.mean(v for v, _ in w) if w else 0.0
        du = statistics.mean(d for _, d in (f or w)) / 1e3
        hbm = 2.0 * fk * 1024 + wk * 1024
        print("%-70s %10d %6d %16.1f %16.1f %12.1f %16.0f" % (k[0][:70], k[1], len(f or w), fk, wk, du, hbm))
        summary["%s|grid=%d" % k] = {"fetch_kb": fk, "write_kb": wk, "avg_us": du, "hbm_bytes_corrected": hbm}
    grid = [int(t) for t in os.environ.get("RML_PMC_GRID", "64x64x128").split("x")]
    doc = {"grid": grid, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 2x FETCH correction for gfx950", "kernels": summary}
    for prefix, key in (("k_project_fast", "project_hbm_bytes_per_launch"), ("k_project_u8_max", "u8_project_hbm_bytes_per_launch"),
                        ("k_project_wave", "walabot_project_hbm_bytes_per_launch")):
        proj = {k: v for k, v in summary.items() if k.startswith(prefix) and v["fetch_kb"] > 1e5}
        if proj:
            bk = max(proj, key=lambda k: proj[k]["fetch_kb"])
            doc[key] = proj[bk]["hbm_bytes_corrected"]
            doc[key.replace("hbm_bytes_per_launch", "frames_per_launch")] = 16384 if prefix == "k_project_wave" else int(bk.split("grid=")[1]) // 256
    if "project_frames_per_launch" in doc:
        doc["frames_per_launch"] = doc["project_frames_per_launch"]
    json.dump(doc, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
