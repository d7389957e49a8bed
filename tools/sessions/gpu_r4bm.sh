#!/bin/bash
# round-4 GPU session BM: DeviceAdam (csrc/optim.hip) -- tests, SGAN step against RML_DEVICE_ADAM=0, bench row, launch count
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=$PWD/gpurun_out/r4bm; mkdir -p $O
R=$PWD
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_gpu.py tests/test_dist_gpu.py -x -q -k "sgan or adam or dist or rccl or bench" 2>&1 | tail -n 4
for k in 0 1 0 1; do echo -n "RML_DEVICE_ADAM=$k "; RML_DEVICE_ADAM=$k timeout 600 python tools/bench_nn.py sgan --steps 100 2>&1 | tail -n 1 | cut -c60-200; done
timeout 600 python bench.py --no-walabot --no-u8 --no-general --no-dnn --no-slice --no-cpu --no-pmc --frames 8192 --steps 3 --warmup 1 > $O/bench_sgan.json 2>$O/bench.err
python - <<PY
import json
d=json.loads([l for l in open("$O/bench_sgan.json") if l.startswith("{")][-1])
print({k:v for k,v in d["doc"]["sgan_train_step"].items() if k in ("value","ms_per_step","c_loss","d_loss","d_fake_loss","replicas_identical")})
PY
cd /tmp
timeout 900 rocprofv3 --kernel-trace -d $O/prof -o k -- python $R/tools/bench_nn.py sgan --steps 40 > $O/run.log 2>&1
DB=$(find $O/prof -name "*.db" | head -1)
python $R/tools/step_gaps.py $DB --marker k_c1_imgstats --per-step 6 --steps 10 > $O/sgan_gaps.txt 2>&1
head -1 $O/sgan_gaps.txt; grep -A 36 "kernel time per step" $O/sgan_gaps.txt | cut -c1-100
rm -rf $O/prof
