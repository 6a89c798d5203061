#!/bin/bash
# SQ counters of the uplift kernels at a large batch (separate --pmc passes, kernel trace only):
#   gpurun -- 'bash tools/pmc_uplift.sh > gpurun_out/pmc_uplift.txt 2>&1'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_uplift; rm -rf $O; mkdir -p $O
cd $R
export TTUP_UPLIFT_B=${TTUP_UPLIFT_B:-2000} TTUP_UPLIFT_T=${TTUP_UPLIFT_T:-120}
python3 tools/bench_uplift.py 2>&1 | grep uplift
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $O/sq1 -- python3 tools/bench_uplift.py > $O/sq1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_WAVES -d $O/sq2 -- python3 tools/bench_uplift.py > $O/sq2.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/sq3 -- python3 tools/bench_uplift.py > $O/sq3.log 2>&1
python3 tools/pmc_summary.py $O/sq1 $O/sq2 $O/sq3 | head -40
tail -3 $O/sq3.log
rm -rf $O
