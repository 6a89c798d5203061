#!/bin/bash
# Quick A/B evidence for one kernel change on the GPU box (through gpurun from the repo root):
#   tools/quick_prof.sh <tag> [pytest -k expression]
# bit-identity / parity tests, isolated launch durations (rocprofv3 --kernel-trace --stats on tools/prof_cnn.py) and the LDS /
# instruction-count PMC pass (separate run).  Output: gpurun_out/<tag>/{tests.log,ktc.csv,pmc.txt}
set -u
TAG=${1:-q}; KEXPR=${2:-"fused_kernels or partially_fused or chain_runtime or bf16_path or fullsize_planted or ragged or special_values"}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $R
python3 -m pytest tests/test_gpu_parity.py -q -x -k "$KEXPR" > $O/tests.log 2>&1; tail -3 $O/tests.log
TTUP_PROF_REPS=20 rocprofv3 --output-format csv --kernel-trace --stats -d $O/ktc -o kt -- python3 tools/prof_cnn.py > $O/ktc.log 2>&1
f=$(find $O/ktc -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $O/ktc.csv
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS -d $O/pmc_sq2 -- python3 tools/prof_cnn.py > $O/pmc_sq2.log 2>&1
python3 tools/pmc_summary.py $O/pmc_sq2 > $O/pmc.txt 2>&1
rm -rf $O/ktc $O/pmc_sq2
python3 - <<PY
import csv
rows=list(csv.DictReader(open('$O/ktc.csv')))
for r in rows[:16]: print('%-70s calls %4s avg_us %9.1f pct %5s'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
grep -i "bb_chain2\|bneck" $O/pmc.txt | head -20
