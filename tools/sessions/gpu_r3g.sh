#!/bin/bash
# round-3 GPU session G: pipeline A/B (half-plane projection buffers beside the GEMM; ring GEMM for uint8 volumes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3g; mkdir -p $O
B="python bench.py --steps 10 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --parity 512"
for rep in 1 2; do
  for v in base half; do
    G=""; [ $v = half ] && G=2
    RML_WAVE_G=$G timeout 900 $B --no-u8 > $O/f32_${v}_$rep.json 2>> $O/f32.err
  done
done
for v in base ring; do
  P=""; [ $v = ring ] && P=1
  RML_PIPE_GEMM=$P timeout 900 $B > $O/u8_$v.json 2>> $O/u8.err
done
for f in $O/f32_*.json $O/u8_*.json; do python tools/exp/show_bench.py $f $(basename $f .json); done
tail -n 3 $O/*.err
