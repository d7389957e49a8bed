#!/bin/bash
# round-3 GPU session AD: ring GEMM with the dead half of the tail SV tile skipped; augmentation throughput leg of the bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3ad; mkdir -p $O
timeout 900 python -m pytest tests/test_svm_gpu.py tests/test_capi_gpu.py -x -q 2>&1 | tail -n 3
timeout 600 python tools/gemm_ab.py exact --grid 64x64x128 --svs 2562 --frames 23808 --rounds 3 2>&1 | grep -v amdgpu.ids | cut -c1-400
timeout 600 python tools/gemm_ab.py exact --grid 22x31x176 --svs 2000 --frames 21760 --rounds 2 2>&1 | grep -v amdgpu.ids | cut -c1-400
timeout 900 python bench.py --steps 10 --warmup 3 --no-dnn --no-sgan --no-cpu --no-pmc --parity 1024 --no-walabot > $O/bench.json 2> $O/bench.err
python tools/exp/show_bench.py $O/bench.json x
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['gemm_roofline']['alone'], d['gemm_roofline']['alone_large_batch']); print(json.dumps(d['general_rows']['augmentation'])[:700]); print(d['general_rows']['value'])"
