#!/bin/bash
# round-3 GPU session H: NN test measurements (-s), the frames-per-launch anomaly of k_project_wave with counters, profile round
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_nn_gpu.py -x -q -s 2>&1 | grep -v "^$" | tail -40 > $O/pytest_nn.txt
( /opt/rocm/bin/rocprofv3 -L 2>/dev/null | grep -i -E "utcl|tlb|translat|GRBM_GUI|TCC_EA0_RDREQ|TCC_HIT|TCC_MISS|TCC_REQ|TCP_TCC_READ|FETCH_SIZE|MemUnit|TCC_TAG_STALL|TCC_BUSY" | head -60 ) > $O/counters_list.txt 2>&1
for n in 2048 4096 8192 12288 16384 24576 32768 65536; do
  timeout 300 python tools/kbench.py proj --grid 22x31x176 --frames $n --iters 20 2>/dev/null | tail -1 >> $O/proj_sweep.jsonl
done
cd /tmp
for n in 8192 16384 32768; do
  for c in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" ; do
    tag=$(echo $c | cut -d' ' -f1)
    timeout 300 rocprofv3 --pmc $c --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_${n}_$tag -o k -- python $GRAFT_REPO_ROOT/tools/kbench.py proj --grid 22x31x176 --frames $n --iters 6 > /dev/null 2>> $GRAFT_REPO_ROOT/$O/pmc.err
    echo "== frames $n counters $c" >> $GRAFT_REPO_ROOT/$O/pmc_sweep.txt
    python $GRAFT_REPO_ROOT/tools/pmc_query.py $GRAFT_REPO_ROOT/$O/pmc_${n}_$tag/k_results.db "%k_project_wave%" >> $GRAFT_REPO_ROOT/$O/pmc_sweep.txt 2>&1
    rm -rf $GRAFT_REPO_ROOT/$O/pmc_${n}_$tag
  done
done
cd $GRAFT_REPO_ROOT
cat $O/pytest_nn.txt; cat $O/counters_list.txt | head -30; cat $O/proj_sweep.jsonl | cut -c1-250; cat $O/pmc_sweep.txt | cut -c1-200; tail -5 $O/pmc.err
