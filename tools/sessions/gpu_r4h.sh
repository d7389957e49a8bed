#!/bin/bash
# round-4 GPU session H: the whole GPU suite; the CNN row with the projection of batch b+1 beside the resize of batch b
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4h; mkdir -p $O
export TMPDIR=/tmp
timeout 3300 python -m pytest tests -q -m gpu 2>&1 | tail -n 15
timeout 900 python bench.py --steps 5 --warmup 2 --no-walabot --no-u8 --no-slice --no-general --no-sgan --no-cpu --no-pmc --parity 256 --frames 8192 > $O/dnn.json 2> $O/dnn.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4h/dnn.json"))
r = d["doc"]["dnn_forward"]
print("dnn", r["value"], r["value_uint8_volumes"], r["roofline"], r["parity"])
PY
