#!/bin/bash
# round-4 GPU session AI (closing): whole GPU suite, smoke, the default bench line, and the CNN-chain parts of the profile round on the final library
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4ai; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
( time timeout 1500 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2800 $O/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dnn -o k -- python $R/bench.py --no-cpu --no-pmc --parity 256 --steps 5 --warmup 2 --frames 8192 --no-walabot --no-u8 --no-slice --no-general --no-sgan > $R/gpurun_out/r04_bench_dnn.json 2> $R/gpurun_out/prof_dnn.err
python $R/tools/prof_summary.py stats $R/gpurun_out/prof_dnn/k_results.db > $R/gpurun_out/r04_stats_dnn.txt
rm -rf $R/gpurun_out/prof_dnn
python - <<PY
import json
d = json.load(open("$R/gpurun_out/r04_bench_dnn.json"))["doc"]["dnn_forward"]
r = d["roofline"]
open("$R/gpurun_out/r04_stats_dnn.txt", "a").write("# bench.py dnn_forward of THIS run: value %.0f frames/s (uint8 volumes %.0f; Pillow-exact resize + float rows in front of the trunk: %.0f, class probabilities differ by <= %.1e); roofline k_dnn_trunk_rf in situ avg %.4f ms (min %.4f max %.4f, %d launches of %d frames) = %.1f TFLOP/s = %.4f of 2500\n"
    % (d["value"], d["value_uint8_volumes"], d["pillow_exact_resize_chain"]["value_this_rank"], d["pillow_exact_resize_chain"]["proba_max_abs_diff_vs_fused_preprocessing"], r["avg_launch_ms"], r["min_launch_ms"], r["max_launch_ms"], r["launches"], r["frames_per_launch"], r["achieved"], r["frac"]))
PY
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace -d $R/gpurun_out/prof_dnn_$c -o k -- python $R/tools/dnn_chain.py --frames 32768 --steps 2 > /dev/null 2> $R/gpurun_out/prof_dnn_$c.err
done
( echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only) on tools/dnn_chain.py --frames 32768 --steps 2 (passes of 16 384 Walabot frames):"
  echo "# per-kernel averages; FETCH_SIZE in KB as reported (x2 = bytes / 1024 on gfx950).  Algorithmic bytes per pass of 16 384 frames: projection 7.87 GB read;"
  echo "# k_pre3 0.164 GB read + 0.629 GB written; k_dnn_trunk_rf 0.629 GB read + 1.258 GB written; k_fc1_splitk 1.258 GB read"
  for c in FETCH_SIZE WRITE_SIZE; do echo "# --- $c"; python $R/tools/pmc_query.py $R/gpurun_out/prof_dnn_$c/k_results.db "%k_%"; done ) > $R/gpurun_out/r04_pmc_dnn.txt 2>&1
rm -rf $R/gpurun_out/prof_dnn_FETCH_SIZE $R/gpurun_out/prof_dnn_WRITE_SIZE
cd $R
head -12 gpurun_out/r04_stats_dnn.txt | cut -c1-160; tail -1 gpurun_out/r04_stats_dnn.txt | cut -c1-400
