#!/usr/bin/env python3
"""Print the headline numbers of a bench.py JSON line: python tools/exp/show_bench.py file.json [tag]"""
import json
import sys

d = json.load(open(sys.argv[1]))
tag = sys.argv[2] if len(sys.argv) > 2 else ""
doc = d.get("doc", d)          # round 4: the verbose rows sit under "doc", the compact ones at the end of the line


def row(name, r):
    if not r:
        return
    u = r.get("uint8_ingest") or {}
    print(tag, "%s f32 %.0f frames/s %.3f ms frac %.3f e2e %.3f |" % (name, r["value"], r["ms_per_step"], r["roofline"]["frac"], r.get("hbm_frac_end_to_end", 0)),
          "u8 %.0f frames/s %.3f ms launch %.4f ms" % (u.get("value", 0), u.get("ms_per_step", 0), (u.get("roofline") or {}).get("avg_launch_ms", 0)))
    for k, s in (r.get("slice_rows") or {}).items():
        print(tag, "   %s %.0f frames/s e2e %.3f kernel %.3f (%.4f ms/launch) gemm chunk %s ms traffic_x %s parity %s" % (
            k, s["value"], s["hbm_frac_end_to_end"], s["roofline"]["frac"], s["roofline"]["avg_launch_ms"],
            (s.get("gemm_roofline") or {}).get("avg_chunk_ms"), s["roofline"].get("traffic_over_algorithmic"), s.get("parity")))


head = dict(doc)
head.update({k: d[k] for k in ("value", "ms_per_step", "hbm_frac_end_to_end") if k in d})
if "doc" in d:
    head["roofline"] = doc["roofline"]
row("headline", head)
row("walabot", doc.get("walabot_grid"))
if "summary" in d:
    print(tag, "gate:", d["summary"].get("parity_gate"))
