#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh rNN
# 1. rocprofv3 --kernel-trace --stats of the default bench workload (fewer steps, no CPU leg)
# 2. two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, kernel trace only -- never with sys/hip/hsa traces)
# 3. kernel stats of the dnn pipeline (tools/bench_nn.py dnn) and of the sgan train step (tools/bench_nn.py sgan)
# Summaries land in gpurun_out/; copy the ones to keep into profiles/.
set -u
TAG=${1:-rXX}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
BENCH="python $R/bench.py --no-cpu --no-pmc --parity 512 --steps 5 --warmup 2"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o k -- $BENCH > $R/gpurun_out/${TAG}_bench_under_rocprofv3.json 2> $R/gpurun_out/prof_stats.err
python $R/tools/prof_summary.py stats $R/gpurun_out/prof_stats/k_results.db > $R/gpurun_out/${TAG}_kernel_stats.txt
# PMC passes over the projection kernels of the fused pipeline's first pass (the same child bench.py measures
# roofline.traffic with: tools/pmc_child.py), one pass per counter, kernel trace only
CFG='[{"tag":"primary_f32","grid":[64,64,128],"frames":16384,"u8":false},{"tag":"primary_u8","grid":[64,64,128],"frames":16384,"u8":true},{"tag":"walabot_f32","grid":[22,31,176],"frames":16384,"u8":false},{"tag":"walabot_u8","grid":[22,31,176],"frames":16384,"u8":true}]'
for c in FETCH_SIZE WRITE_SIZE; do
    RML_WAVE_SHARE=1 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/prof_$c -o k -- python $R/tools/pmc_child.py "$CFG" > /dev/null 2> $R/gpurun_out/prof_$c.err
done
cd $R
python tools/prof_summary.py pmc gpurun_out/prof_FETCH_SIZE/k_results.db gpurun_out/prof_WRITE_SIZE/k_results.db > gpurun_out/${TAG}_pmc.txt
cp profiles/pmc_latest.json gpurun_out/pmc_latest.json
# matrix-core / issue counters of the exact GEMMs alone (both tile sizes) and FETCH of the large-tile kernel
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/prof_mfma -o k -- python $R/tools/kbench.py gemm --grid 64x64x128 --frames 16384 --svs 2562 --iters 6 > /dev/null 2> $R/gpurun_out/prof_mfma.err
RML_GEMM_BIG=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/prof_mfma_small -o k -- python $R/tools/kbench.py gemm --grid 64x64x128 --frames 16384 --svs 2562 --iters 6 > /dev/null 2>> $R/gpurun_out/prof_mfma.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_gemm_fetch -o k -- python $R/tools/kbench.py gemm --grid 64x64x128 --frames 16384 --svs 2562 --iters 6 > /dev/null 2>> $R/gpurun_out/prof_mfma.err
RML_GEMM_BIG=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_gemm_fetch_small -o k -- python $R/tools/kbench.py gemm --grid 64x64x128 --frames 16384 --svs 2562 --iters 6 > /dev/null 2>> $R/gpurun_out/prof_mfma.err
( echo "# rocprofv3 --pmc (kernel trace only) on tools/kbench.py gemm --grid 64x64x128 --frames 16384 --svs 2562: per-kernel averages"; echo "# --- k_svm_gemm_i8_256 (default for large batches)"; python $R/tools/pmc_query.py $R/gpurun_out/prof_mfma/k_results.db "%svm_gemm%"; echo "# --- k_svm_gemm<I8> 128x128 (RML_GEMM_BIG=0)"; python $R/tools/pmc_query.py $R/gpurun_out/prof_mfma_small/k_results.db "%svm_gemm%"; echo "# --- FETCH_SIZE (KB as reported; x2 = bytes / 1024 on gfx950)"; python $R/tools/pmc_query.py $R/gpurun_out/prof_gemm_fetch/k_results.db "%svm_gemm%"; python $R/tools/pmc_query.py $R/gpurun_out/prof_gemm_fetch_small/k_results.db "%svm_gemm%" ) > $R/gpurun_out/${TAG}_pmc_gemm.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_nn -o k -- python $R/tools/bench_nn.py dnn --frames 16384 > $R/gpurun_out/${TAG}_bench_nn_dnn.json 2> $R/gpurun_out/prof_nn.err
python $R/tools/prof_summary.py stats $R/gpurun_out/prof_nn/k_results.db > $R/gpurun_out/${TAG}_kernel_stats_dnn.txt
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sgan -o k -- python $R/tools/bench_nn.py sgan --steps 100 > $R/gpurun_out/${TAG}_bench_nn_sgan.json 2> $R/gpurun_out/prof_sgan.err
python $R/tools/prof_summary.py stats $R/gpurun_out/prof_sgan/k_results.db > $R/gpurun_out/${TAG}_kernel_stats_sgan.txt
cd $R
rm -rf gpurun_out/prof_stats gpurun_out/prof_FETCH_SIZE gpurun_out/prof_WRITE_SIZE gpurun_out/prof_nn gpurun_out/prof_sgan gpurun_out/prof_mfma gpurun_out/prof_mfma_small gpurun_out/prof_gemm_fetch gpurun_out/prof_gemm_fetch_small
head -12 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-160
head -12 gpurun_out/${TAG}_pmc.txt | cut -c1-170
cat gpurun_out/${TAG}_pmc_gemm.txt | cut -c1-170
head -8 gpurun_out/${TAG}_kernel_stats_dnn.txt | cut -c1-160
