#!/bin/bash
# round-4 GPU session P: per-workload profiles of the round (tools/profile_round.sh r04) and the full bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
bash tools/profile_round.sh r04 2>&1 | tail -n 80
timeout 1500 python bench.py > gpurun_out/r04_bench_1gpu.json 2> gpurun_out/r04_bench_1gpu.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/r04_bench_1gpu.json; echo
python tools/exp/show_bench.py gpurun_out/r04_bench_1gpu.json final
