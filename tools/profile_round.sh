#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):   tools/profile_round.sh rNN
# ONE rocprofv3 --kernel-trace --stats run PER WORKLOAD, so that every roofline.frac of the bench line can be recomputed from
# one row of one summary (a summary over the whole bench mixes launches of one kernel on different shapes):
#   stats_headline_f32   64x64x128 f32 volumes -> SVM          (k_project_wave + k_svm_gemm<I8> sharing the CUs)
#   stats_walabot_f32    22x31x176 f32 volumes -> SVM          (k_project_lin + k_svm_gemm<I8> sharing the CUs)
#   stats_headline_u8 / stats_walabot_u8    the same frames as uint8 volumes (k_project_u8_max, then k_svm_gemm_ring<PT,0> in whole rounds)
#   stats_general_rows   rows off the code grid -> multi-digit int8 kernel (k_svm_gemm_ring<PT,1>) and float64 MFMA
#   stats_gemm_alone     code rows -> k_svm_gemm_ring<PT,0>, whole-round chunks (tools/gemm_ab.py)
#   stats_dnn            bench.py's own CNN row (the line's dnn_forward.roofline and this summary's k_dnn_trunk_rf row come from ONE run)
#   stats_sgan
#   stats_latency        tools/latency.py: one observation per call (k_project_finalize, k_svm_dot_small, k_svm_epi_small, k_svm_gemm_splitk)
#   stats_slice          the reference-faithful rows (k_derive_slice, k_slice_rows) cut out of the two f32 runs, both grids
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
    # bench.py prints two lines (verbose rows, contract line): its rows are read from the --doc-file JSON ({"doc": ..., "line": ...})
    local doc=()
    case "$*" in *bench.py*) doc=(--doc-file $R/gpurun_out/${TAG}_bench_$name.json);; esac
    rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$name -o k -- "$@" "${doc[@]}" > $R/gpurun_out/${TAG}_bench_$name.log 2> $R/gpurun_out/prof_$name.err
    python $R/tools/prof_summary.py stats $R/gpurun_out/prof_$name/k_results.db > $R/gpurun_out/${TAG}_stats_$name.txt
    # kernel timeline of the steady state of the two float32 pipelines (which stream carries what, the gaps of the projection stream)
    case $name in
        headline_f32) python $R/tools/timeline.py $R/gpurun_out/prof_$name/k_results.db --rows 50 > $R/gpurun_out/${TAG}_timeline_headline.txt 2>&1;;
        walabot_f32) python $R/tools/timeline.py $R/gpurun_out/prof_$name/k_results.db --rows 50 > $R/gpurun_out/${TAG}_timeline_walabot.txt 2>&1;;
    esac
    rm -rf $R/gpurun_out/prof_$name
}
prof headline_f32 python $R/bench.py $COMMON --no-walabot --no-u8 --no-general --no-dnn --no-sgan
prof walabot_f32 python $R/bench.py $COMMON --grid 22x31x176 --frames 262144 --no-u8 --no-general --no-dnn --no-sgan
prof headline_u8 python $R/bench.py $COMMON --ingest u8 --no-walabot --no-general --no-dnn --no-sgan
prof walabot_u8 python $R/bench.py $COMMON --ingest u8 --grid 22x31x176 --frames 262144 --no-general --no-dnn --no-sgan
prof general_rows python $R/bench.py $COMMON --frames 8192 --no-walabot --no-u8 --no-dnn --no-sgan
prof gemm_alone python $R/tools/gemm_ab.py exact --grid 64x64x128 --svs 2562 --frames 23808 --rounds 2
prof dnn python $R/bench.py --no-cpu --no-pmc --parity 256 --steps 5 --warmup 2 --frames 8192 --no-walabot --no-u8 --no-slice --no-general --no-sgan
python - <<PY
import json
d = json.load(open("$R/gpurun_out/${TAG}_bench_dnn.json"))["doc"]["dnn_forward"]
r = d["roofline"]
open("$R/gpurun_out/${TAG}_stats_dnn.txt", "a").write("# bench.py dnn_forward of THIS run: value %.0f frames/s (uint8 volumes %.0f); roofline k_dnn_trunk_rf in situ avg %.4f ms (min %.4f max %.4f, %d launches of %d frames) = %.1f TFLOP/s = %.4f of 2500\n"
    % (d["value"], d["value_uint8_volumes"], r["avg_launch_ms"], r["min_launch_ms"], r["max_launch_ms"], r["launches"], r["frames_per_launch"], r["achieved"], r["frac"]))
PY
prof sgan python $R/tools/bench_nn.py sgan --steps 100
# one observation per call: the kernels of a B = 1 call, and the host-clock latencies of both grids
prof latency python $R/tools/latency.py --grid 64x64x128 --svs 2560 --iters 30
( for g in "64x64x128 2560" "22x31x176 2000"; do set -- $g; python $R/tools/latency.py --grid $1 --svs $2 2>/dev/null | tail -1; done ) > $R/gpurun_out/${TAG}_latency.txt
( echo "# k_derive_slice / k_slice_rows rows of ${TAG}_stats_headline_f32.txt (64x64x128) and ${TAG}_stats_walabot_f32.txt (22x31x176):"
  echo "# bench.py's slice_rows.derive_slice_svm / slice_mode (doc) of the same runs: in-situ avg launch ms beside them"
  for n in headline_f32 walabot_f32; do
    head -2 $R/gpurun_out/${TAG}_stats_$n.txt | tail -1
    grep -E "k_derive_slice|k_slice_rows" $R/gpurun_out/${TAG}_stats_$n.txt
    python - <<PY
import json
d = json.load(open("$R/gpurun_out/${TAG}_bench_$n.json"))["doc"]
for k, r in (d.get("slice_rows") or {}).items():
    print("#   %s %s: %.0f frames/s, e2e %.4f of 8 TB/s, %s in situ %.4f ms per %d frames = %.4f" % ("$n", k, r["value"], r["hbm_frac_end_to_end"], r["roofline"]["kernel"], r["roofline"]["avg_launch_ms"], r["roofline"]["frames_per_launch"], r["roofline"]["frac"]))
PY
  done ) > $R/gpurun_out/${TAG}_stats_slice.txt 2>&1
# PMC passes over the projection kernels of the fused pipeline's first pass (the same child bench.py measures
# roofline.traffic with: tools/pmc_child.py), one pass per counter, kernel trace only
CFG='[{"tag":"primary_f32","grid":[64,64,128],"frames":8192,"u8":false},{"tag":"primary_u8","grid":[64,64,128],"frames":8192,"u8":true},{"tag":"walabot_f32","grid":[22,31,176],"frames":8192,"u8":false},{"tag":"walabot_u8","grid":[22,31,176],"frames":16384,"u8":true},{"tag":"primary_derive","grid":[64,64,128],"frames":8192,"u8":false,"mode":"derive_slice"},{"tag":"primary_slice","grid":[64,64,128],"frames":8192,"u8":false,"mode":"slice"},{"tag":"walabot_derive","grid":[22,31,176],"frames":8192,"u8":false,"mode":"derive_slice"},{"tag":"walabot_slice","grid":[22,31,176],"frames":8192,"u8":false,"mode":"slice"}]'
for c in FETCH_SIZE WRITE_SIZE; do
    RML_WAVE_SHARE=1 rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/prof_$c -o k -- python $R/tools/pmc_child.py "$CFG" > /dev/null 2> $R/gpurun_out/prof_$c.err
done
# the CNN chain's kernels (configs[3]: code-row projection, k_pre3, trunk, k_fc1_splitk): FETCH_SIZE / WRITE_SIZE per launch
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/prof_dnn_$c -o k -- python $R/tools/dnn_chain.py --frames 32768 --steps 2 > /dev/null 2> $R/gpurun_out/prof_dnn_$c.err
done
( echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only) on tools/dnn_chain.py --frames 32768 --steps 2 (passes of 16 384 Walabot frames):"
  echo "# per-kernel averages; FETCH_SIZE in KB as reported (x2 = bytes / 1024 on gfx950).  Algorithmic bytes per pass of 16 384 frames: projection 7.87 GB read;"
  echo "# k_pre3 0.164 GB read + 0.629 GB written; k_dnn_trunk_rf 0.629 GB read + 1.258 GB written; k_fc1_splitk 1.258 GB read"
  for c in FETCH_SIZE WRITE_SIZE; do echo "# --- $c"; python $R/tools/pmc_query.py $R/gpurun_out/prof_dnn_$c/k_results.db "%k_%"; done ) > $R/gpurun_out/${TAG}_pmc_dnn.txt 2>&1
rm -rf $R/gpurun_out/prof_dnn_FETCH_SIZE $R/gpurun_out/prof_dnn_WRITE_SIZE
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
# chip clock under the CNN trunk: GRBM_GUI_ACTIVE / duration (the in-situ and the rocprofv3 duration of k_dnn_trunk_rf agree only at the same clock)
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d $R/gpurun_out/prof_trunk -o k -- python $R/tools/trunk_bench.py --reps 3 > /dev/null 2> $R/gpurun_out/prof_trunk.err
( echo "# rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES on tools/trunk_bench.py (8192 samples of 3x80x80 bf16): clock = GRBM_GUI_ACTIVE / dur"
  python $R/tools/pmc_query.py $R/gpurun_out/prof_trunk/k_results.db "%trunk%" ) > $R/gpurun_out/${TAG}_pmc_trunk.txt 2>&1
cd $R
rm -rf gpurun_out/prof_trunk
rm -rf gpurun_out/prof_FETCH_SIZE gpurun_out/prof_WRITE_SIZE gpurun_out/prof_mfma gpurun_out/prof_mfma_small gpurun_out/prof_gemm_fetch gpurun_out/prof_gemm_fetch_small
for f in gpurun_out/${TAG}_stats_*.txt; do echo "== $f"; head -6 $f | cut -c1-170; done
head -16 gpurun_out/${TAG}_pmc.txt | cut -c1-170
cat gpurun_out/${TAG}_stats_slice.txt | cut -c1-200
cat gpurun_out/${TAG}_pmc_trunk.txt | cut -c1-170
cat gpurun_out/${TAG}_pmc_gemm.txt | cut -c1-170
