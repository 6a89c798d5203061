#!/bin/bash
# Small-batch uplift latency with and without the stage kernel, and the throughput at growing batch sizes for the workgroup limit.
#   gpurun -- 'bash tools/uplift_latency.sh > gpurun_out/uplift_latency.txt 2>&1'
cd "$(dirname "$0")/.."
for cfg in "1 50" "1 48" "3 121" "1 20" "16 50" "64 50"; do
  set -- $cfg
  echo "== B=$1 T=$2"
  TTUP_UPLIFT_B=$1 TTUP_UPLIFT_T=$2 python tools/bench_uplift.py | tail -1
  TTUP_UPLIFT_NO_STAGE=1 TTUP_UPLIFT_B=$1 TTUP_UPLIFT_T=$2 python tools/bench_uplift.py | tail -1
done
for b in 256 1024 4096; do
  for wg in 0 256 1024 100000; do
    echo "== B=$b T=50 stage_wg=$wg"; TTUP_UPLIFT_STAGE_WG=$wg TTUP_UPLIFT_B=$b TTUP_UPLIFT_T=50 python tools/bench_uplift.py | head -1
  done
done
