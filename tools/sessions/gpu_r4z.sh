#!/bin/bash
# round-4 GPU session Z: LDS code stage of the wave-per-frame kernels (codes-only launches): tests, A/B alone and in the pipelines
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4z; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_projection_gpu.py -x -q -k "code_stage or linear_plane or wave" 2>&1 | tail -n 5
for rep in 1 2; do
  for k in 0 1; do
    printf "stage=%s walabot  " $k; RML_STAGE_CODES=$k timeout 300 python tools/kbench.py proj --grid 22x31x176 --frames 16384 | grep "codes+stats only" | cut -c75-200
    printf "stage=%s 64x64x128 (pipeline config) " $k; RML_WAVE_SHARE=1 RML_STAGE_CODES=$k timeout 300 python tools/kbench.py proj --grid 64x64x128 --frames 8192 | grep "codes+stats only" | cut -c75-200
  done
done
B="python bench.py --steps 8 --warmup 3 --no-general --no-dnn --no-sgan --no-cpu --no-pmc --no-u8 --no-slice --parity 1024"
for rep in 1 2; do for k in 0 1; do
  RML_STAGE_CODES=$k timeout 900 $B > $O/stage${k}_$rep.json 2>> $O/b.err
  python tools/exp/show_bench.py $O/stage${k}_$rep.json stage$k | cut -c1-150
done; done
