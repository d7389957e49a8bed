#!/bin/bash
# round-3 GPU session R: k_project_u8_max with the cross-lane steps on the VALU (permlane swaps / DPP) and the byte epilogue
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3r; mkdir -p $O
timeout 900 python -m pytest tests/test_projection_gpu.py tests/test_capi_gpu.py -x -q 2>&1 | tail -n 5
for x in 1 0; do
  echo "== RML_U8_XLANE=$x"
  RML_U8_XLANE=$x timeout 300 python tools/kbench.py project --u8 --grid 64x64x128 --frames 8192 2>/dev/null | cut -c1-300
  RML_U8_XLANE=$x timeout 300 python tools/kbench.py project --u8 --grid 22x31x176 --frames 16384 2>/dev/null | cut -c1-300
done
B="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 1024 --ingest u8"
for rep in 1 2; do for x in 1 0; do
  RML_U8_XLANE=$x timeout 900 $B > $O/x${x}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/x${x}_$rep.json xlane$x
done; done
