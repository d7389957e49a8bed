cd /tmp; export TMPDIR=/tmp; R=/root/repo
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/ptrunk_$tag -o k -- python $R/tools/trunk_bench.py --reps 3 > /dev/null 2> $R/gpurun_out/ptrunk_$tag.err
  python $R/tools/pmc_query.py $R/gpurun_out/ptrunk_$tag/k_results.db "%trunk%"
done
