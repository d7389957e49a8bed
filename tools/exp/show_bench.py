#!/usr/bin/env python3
"""Print the headline numbers of a bench.py JSON line: python tools/exp/show_bench.py file.json [tag]"""
import json
import sys

d = json.load(open(sys.argv[1]))
tag = sys.argv[2] if len(sys.argv) > 2 else ""
u = d.get("uint8_ingest") or {}
print(tag, "f32 %.0f frames/s %.3f ms frac %.3f |" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]),
      "u8 %.0f frames/s %.3f ms launch %.4f ms" % (u.get("value", 0), u.get("ms_per_step", 0), (u.get("roofline") or {}).get("avg_launch_ms", 0)))
w = d.get("walabot_grid")
if w:
    u = w.get("uint8_ingest") or {}
    print(tag, "walabot f32 %.0f frames/s frac %.3f |" % (w["value"], w["roofline"]["frac"]),
          "u8 %.0f frames/s %.3f ms launch %.4f ms" % (u.get("value", 0), u.get("ms_per_step", 0), (u.get("roofline") or {}).get("avg_launch_ms", 0)))
