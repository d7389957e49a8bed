#!/bin/bash
# round-3 GPU session V: SGAN graph step that writes its gradients (no accumulate kernels); full bench line again (cpu_baseline + traffic legs)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3v; mkdir -p $O
timeout 900 python -m pytest tests/test_nn_gpu.py tests/test_dist_gpu.py -x -q 2>&1 | tail -n 8
timeout 600 python tools/bench_nn.py sgan --steps 100 2>&1 | grep -v amdgpu.ids | cut -c1-300
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python tools/exp/show_bench.py $O/bench.json full; tail -n 3 $O/bench.err
python -c "
import json; d=json.load(open('$O/bench.json')); print(json.dumps(d['cpu_baseline'])[:400]); print(d['roofline']['traffic'], json.dumps(d['sgan_train_step'])[:300])"
