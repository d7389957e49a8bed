#!/usr/bin/env python3
"""Per-kernel PMC counters of an arbitrary command: one `rocprofv3 --kernel-trace --pmc <counter>` pass per counter (kernel
trace only, as the pool requires), then the mean counter value and duration per kernel name (first dispatch of a name = warm-up).

    python tools/pmc_kernels.py --match 'k_slice_rows|k_derive_slice' --counters FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum -- \
        python tools/kbench.py slice --grid 64x64x128 --frames 16384 --iters 3

FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; on gfx950 FETCH_SIZE tallies 128-byte requests of a wide coalesced
stream at 64 B (MI355X_MICROARCH.md, HBM section) -- the raw value is printed, and `fetch_x2_bytes` beside it.
"""
import argparse
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_support import _find_db, _read_counter      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--match", default=".")
    ap.add_argument("--counters", nargs="+", default=["FETCH_SIZE", "WRITE_SIZE"])
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    rx = re.compile(a.match)
    cwd = os.getcwd()
    cmd = [os.path.join(cwd, c) if (c.endswith(".py") and not os.path.isabs(c)) else c for c in cmd]
    res = collections.OrderedDict()
    tmp = tempfile.mkdtemp(prefix="rml_pmck_", dir="/tmp")
    try:
        for counter in a.counters:
            out = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "-d", out, "-o", "k", "--"] + cmd, cwd="/tmp",
                               env=dict(os.environ, TMPDIR="/tmp", PYTHONPATH=cwd), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            if r.returncode != 0:
                print(json.dumps({"counter": counter, "error": r.stderr.decode(errors="replace")[-400:]}))
                continue
            db = _find_db(out)
            rows = _read_counter(db, counter) if db else []
            per = collections.OrderedDict()
            for name, val, dur in rows:
                if rx.search(name):
                    per.setdefault(name, []).append((val, dur))
            for name, vs in per.items():
                vs = vs[1:] if len(vs) > 1 else vs
                short = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
                e = res.setdefault(short, {"dispatches": len(vs)})
                e[counter] = sum(v for v, _ in vs) / len(vs)
                e["avg_us"] = round(sum(d for _, d in vs) / len(vs) / 1e3, 1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for k, e in res.items():
        if "FETCH_SIZE" in e:
            e["fetch_x2_bytes"] = 2048.0 * e["FETCH_SIZE"]
        if "WRITE_SIZE" in e:
            e["write_bytes"] = 1024.0 * e["WRITE_SIZE"]
        print(json.dumps(dict(kernel=k, **e)))


if __name__ == "__main__":
    main()
