cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp; R=$PWD; O=gpurun_out/r3ap; mkdir -p $O; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $R/$O/p1 -o k -- python $R/tools/bench_nn.py dnn --frames 16384 > /dev/null 2> $R/$O/p1.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace -d $R/$O/p2 -o k -- python $R/tools/bench_nn.py dnn --frames 16384 > /dev/null 2> $R/$O/p2.err
cd $R
python tools/pmc_query.py $O/p1/k_results.db "%k_resize%"; python tools/pmc_query.py $O/p2/k_results.db "%k_resize%"
rm -rf $O/p1 $O/p2
