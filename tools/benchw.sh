#!/bin/bash
# usage: benchw.sh  (env passes through) -> compact summary of the Walabot-grid workload
python bench.py --grid 22x31x176 --frames 262144 --no-dnn --no-sgan --no-cpu --no-u8 --parity 512 --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('value',d['value'],'ms',d['ms_per_step'],'e2e',d['hbm_frac_end_to_end'],'proj_frac',d['roofline']['frac'],'proj_ms',d['roofline']['avg_launch_ms'],'fpl',d['roofline']['frames_per_launch'],'gemm_ms',d['gemm_roofline']['avg_chunk_ms'],'alone',d['gemm_roofline']['alone']['ms'],'mism',d['parity']['label_calib_mismatch'])
"
