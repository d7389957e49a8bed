#!/bin/bash
# round-5 GPU session T: LDS counters of the exact GEMM kernels alone (review r4 item 7 asked for the bank-conflict counters beside MFMA busy)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; O=gpurun_out/r5t; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
GEMM="python $R/tools/kbench.py gemm --grid 64x64x128 --frames 23808 --svs 2562 --iters 6"
CNT="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
RML_CHUNK=23808 rocprofv3 --pmc $CNT --kernel-trace -d $R/$O/prof_lds -o k -- $GEMM > /dev/null 2> $R/$O/lds.err
RML_CHUNK=23808 RML_GEMM_BIG=0 rocprofv3 --pmc $CNT --kernel-trace -d $R/$O/prof_lds_small -o k -- $GEMM > /dev/null 2>> $R/$O/lds.err
( echo "# --- LDS counters (session r5t): rocprofv3 --pmc $CNT on the same command; ring 256x256, then 128x128"
  python $R/tools/pmc_query.py $R/$O/prof_lds/k_results.db "%svm_gemm%"
  python $R/tools/pmc_query.py $R/$O/prof_lds_small/k_results.db "%svm_gemm%" ) > $R/$O/pmc_gemm_lds.txt 2>&1
rm -rf $R/$O/prof_lds $R/$O/prof_lds_small
cat $R/$O/pmc_gemm_lds.txt | cut -c1-160; tail -3 $R/$O/lds.err
