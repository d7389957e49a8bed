#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):   tools/profile_round.sh rNN
# ONE rocprofv3 --kernel-trace --stats run PER WORKLOAD, so that every roofline.frac of the bench line can be recomputed from
# one row of one summary (a summary over the whole bench mixes launches of one kernel on different shapes):
#   stats_headline_f32   64x64x128 f32 volumes -> SVM          (k_project_wave + k_svm_gemm<I8> sharing the CUs)
#   stats_walabot_f32    22x31x176 f32 volumes -> SVM          (k_project_lin + k_svm_gemm<I8> sharing the CUs)
#   stats_headline_u8 / stats_walabot_u8    the same frames as uint8 volumes (k_project_u8_max, then k_svm_gemm_ring<PT,0> in whole rounds)
#   stats_general_rows   rows off the code grid -> multi-digit int8 kernel (k_svm_gemm_ring<PT,1>) and float64 MFMA
#   stats_gemm_alone     code rows -> k_svm_gemm_ring<PT,0>, whole-round chunks (tools/gemm_ab.py)
#   stats_dnn / stats_sgan
# then PMC passes (separate runs, kernel trace only -- never with sys/hip/hsa traces): FETCH_SIZE / WRITE_SIZE of the projection
# kernels, matrix-core / issue counters and FETCH_SIZE of the GEMM kernels.  Summaries land in gpurun_out/; copy the ones to
# keep into profiles/.
set -u
TAG=${1:-rXX}
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
COMMON="--no-cpu --no-pmc --parity 512 --steps 5 --warmup 2"
prof() {   # prof <name> <command...>
    local name=$1; shift
    rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$name -o k -- "$@" > $R/gpurun_out/${TAG}_bench_$name.json 2> $R/gpurun_out/prof_$name.err
    python $R/tools/prof_summary.py stats $R/gpurun_out/prof_$name/k_results.db > $R/gpurun_out/${TAG}_stats_$name.txt
    rm -rf $R/gpurun_out/prof_$name
}
prof headline_f32 python $R/bench.py $COMMON --no-walabot --no-u8 --no-general --no-dnn --no-sgan
prof walabot_f32 python $R/bench.py $COMMON --grid 22x31x176 --frames 262144 --no-u8 --no-general --no-dnn --no-sgan
prof headline_u8 python $R/bench.py $COMMON --ingest u8 --no-walabot --no-general --no-dnn --no-sgan
prof walabot_u8 python $R/bench.py $COMMON --ingest u8 --grid 22x31x176 --frames 262144 --no-general --no-dnn --no-sgan
prof general_rows python $R/bench.py $COMMON --frames 8192 --no-walabot --no-u8 --no-dnn --no-sgan
prof gemm_alone python $R/tools/gemm_ab.py exact --grid 64x64x128 --svs 2562 --frames 23808 --rounds 2
prof dnn python $R/tools/bench_nn.py dnn --frames 16384
prof sgan python $R/tools/bench_nn.py sgan --steps 100
# PMC passes over the projection kernels of the fused pipeline's first pass (the same child bench.py measures
# roofline.traffic with: tools/pmc_child.py), one pass per counter, kernel trace only
CFG='[{"tag":"primary_f32","grid":[64,64,128],"frames":8192,"u8":false},{"tag":"primary_u8","grid":[64,64,128],"frames":8192,"u8":true},{"tag":"walabot_f32","grid":[22,31,176],"frames":8192,"u8":false},{"tag":"walabot_u8","grid":[22,31,176],"frames":16384,"u8":true}]'
for c in FETCH_SIZE WRITE_SIZE; do
    RML_WAVE_SHARE=1 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/prof_$c -o k -- python $R/tools/pmc_child.py "$CFG" > /dev/null 2> $R/gpurun_out/prof_$c.err
done
cd $R
python tools/prof_summary.py pmc gpurun_out/prof_FETCH_SIZE/k_results.db gpurun_out/prof_WRITE_SIZE/k_results.db > gpurun_out/${TAG}_pmc.txt
cp profiles/pmc_latest.json gpurun_out/pmc_latest.json 2>/dev/null
# matrix-core / issue counters of the exact GEMMs alone (ring 256x256, 128x128) and their FETCH
cd /tmp
GEMM="python $R/tools/kbench.py gemm --grid 64x64x128 --frames 23808 --svs 2562 --iters 6"
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE"
RML_CHUNK=23808 rocprofv3 --pmc $CNT --kernel-trace -d $R/gpurun_out/prof_mfma -o k -- $GEMM > /dev/null 2> $R/gpurun_out/prof_mfma.err
RML_CHUNK=23808 RML_GEMM_BIG=0 rocprofv3 --pmc $CNT --kernel-trace -d $R/gpurun_out/prof_mfma_small -o k -- $GEMM > /dev/null 2>> $R/gpurun_out/prof_mfma.err
RML_CHUNK=23808 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_gemm_fetch -o k -- $GEMM > /dev/null 2>> $R/gpurun_out/prof_mfma.err
RML_CHUNK=23808 RML_GEMM_BIG=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_gemm_fetch_small -o k -- $GEMM > /dev/null 2>> $R/gpurun_out/prof_mfma.err
( echo "# rocprofv3 --pmc (kernel trace only) on tools/kbench.py gemm --grid 64x64x128 --frames 23808 --svs 2562 (one chunk): per-kernel averages"
  echo "# --- k_svm_gemm_ring<PT,0> (256x256, 5-slot ring: the default for large batches)"; python $R/tools/pmc_query.py $R/gpurun_out/prof_mfma/k_results.db "%svm_gemm%"
  echo "# --- k_svm_gemm<I8> 128x128 (RML_GEMM_BIG=0)"; python $R/tools/pmc_query.py $R/gpurun_out/prof_mfma_small/k_results.db "%svm_gemm%"
  echo "# --- FETCH_SIZE (KB as reported; x2 = bytes / 1024 on gfx950): ring, then 128x128"
  python $R/tools/pmc_query.py $R/gpurun_out/prof_gemm_fetch/k_results.db "%svm_gemm%"; python $R/tools/pmc_query.py $R/gpurun_out/prof_gemm_fetch_small/k_results.db "%svm_gemm%" ) > $R/gpurun_out/${TAG}_pmc_gemm.txt 2>&1
cd $R
rm -rf gpurun_out/prof_FETCH_SIZE gpurun_out/prof_WRITE_SIZE gpurun_out/prof_mfma gpurun_out/prof_mfma_small gpurun_out/prof_gemm_fetch gpurun_out/prof_gemm_fetch_small
for f in gpurun_out/${TAG}_stats_*.txt; do echo "== $f"; head -6 $f | cut -c1-170; done
head -12 gpurun_out/${TAG}_pmc.txt | cut -c1-170
cat gpurun_out/${TAG}_pmc_gemm.txt | cut -c1-170
