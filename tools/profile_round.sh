#!/bin/bash
# Regenerates the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r2 <commit>
# kernel traces of bench.py (two lanes = the timed configuration, one lane = isolated launch durations) and separate PMC
# passes (never combined with a trace domain other than --kernel-trace) on tools/prof_cnn.py / tools/prof_argmax.py.
# Raw output: gpurun_out/<tag>/...; summaries: gpurun_out/<tag>/summary/ (copied into profiles/ by the caller).
set -u
TAG=${1:-r3}; COMMIT=${2:-unknown}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; S=$O/summary
rm -rf $O; mkdir -p $S
cd $R
rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt2 -o kt -- python3 bench.py --no-cpu-baseline --no-extras --steps 4 > $S/${TAG}_bench_lanes2.json 2> $O/kt2.err
# one lane and no fp32 crop passes: only the side streams (pre-processing of the next step, refine, uplift) still overlap
TTUP_LANES=1 rocprofv3 --output-format csv --kernel-trace --stats -d $O/kt1 -o kt -- python3 bench.py --no-cpu-baseline --no-extras --no-certify --steps 4 > $S/${TAG}_bench_lanes1.json 2> $O/kt1.err
rocprofv3 --output-format csv --kernel-trace --stats -d $O/ktu -o kt -- python3 tools/bench_uplift.py > $O/ktu.log 2>&1
# one micro-batch of the CNN alone on one stream, 20 repeats: nothing else runs, the averages are isolated launch durations
TTUP_PROF_REPS=20 rocprofv3 --output-format csv --kernel-trace --stats -d $O/ktc -o kt -- python3 tools/prof_cnn.py > $O/ktc.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -- python3 tools/prof_cnn.py > $O/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -- python3 tools/prof_cnn.py > $O/pmc_write.log 2>&1
# check of the FETCH_SIZE x 2 correction on these kernels' own access patterns: read requests by size class and DRAM reads in 32-B units
rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B -d $O/pmc_rq -- python3 tools/prof_cnn.py > $O/pmc_rq.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_32B TCC_EA0_RDREQ_DRAM TCC_HIT TCC_MISS -d $O/pmc_dram -- python3 tools/prof_cnn.py > $O/pmc_dram.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_argmax -- python3 tools/prof_argmax.py > $O/pmc_fetch_argmax.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_argmax -- python3 tools/prof_argmax.py > $O/pmc_write_argmax.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU -d $O/pmc_sq1 -- python3 tools/prof_cnn.py > $O/pmc_sq1.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS -d $O/pmc_sq2 -- python3 tools/prof_cnn.py > $O/pmc_sq2.log 2>&1
for d in kt2 kt1 ktu ktc; do f=$(find $O/$d -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && { echo "# commit $COMMIT; $(date -u +%FT%TZ); rocprofv3 --output-format csv --kernel-trace --stats ($d)" > $S/${TAG}_kernel_stats_$d.csv; cat $f >> $S/${TAG}_kernel_stats_$d.csv; }; done
python3 tools/pmc_reqsizes.py $O/pmc_rq $O/pmc_fetch > $S/${TAG}_reqsizes.json; python3 tools/pmc_reqsizes.py $O/pmc_dram > $S/${TAG}_dram_reads.json
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_fetch_argmax $O/pmc_write_argmax > $S/${TAG}_traffic.json
{ echo "# commit $COMMIT; $(date -u +%FT%TZ); rocprofv3 --pmc passes (separate runs) on tools/prof_cnn.py: counter averages per launch"; python3 tools/pmc_summary.py $O/pmc_sq1 $O/pmc_sq2; } > $S/${TAG}_pmc_summary.txt
ls -la $S
