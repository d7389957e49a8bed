#!/bin/bash
# round-3 GPU session U: full -m gpu suite, per-workload profile round, full default bench line (state after the uint8 kernel / pipeline changes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3u; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 5 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
timeout 300 python tools/kbench.py proj --grid 64x64x256 --frames 2048 2>&1 | grep -v amdgpu.ids | cut -c1-220
timeout 300 python tools/kbench.py proj --u8 --grid 64x64x256 --frames 4096 2>&1 | grep -v amdgpu.ids | cut -c1-220
timeout 1500 bash tools/profile_round.sh r03 > $O/profile_round.log 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python tools/exp/show_bench.py $O/bench.json full; tail -n 3 $O/bench.err
