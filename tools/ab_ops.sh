#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by +-4 %): per-kernel timings of one CNN micro-batch (tools/ops_report.py,
# HIP events in graph order), the builds alternating.   tools/ab_ops.sh <libA.so|default> <libB.so|default> [rounds]
# Output: gpurun_out/ab/{A,B}_<round>.log and a summary of the per-kernel medians.
A=${1:-default}; B=${2:-default}; N=${3:-3}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
for r in $(seq 1 $N); do
  for v in A B; do
    lib=$A; [ $v = B ] && lib=$B
    if [ "$lib" = default ]; then env -u TTUP_LIB TTUP_REPS=${TTUP_REPS:-10} python3 tools/ops_report.py > gpurun_out/ab/${v}_$r.log 2>&1
    else TTUP_LIB=$lib TTUP_REPS=${TTUP_REPS:-10} python3 tools/ops_report.py > gpurun_out/ab/${v}_$r.log 2>&1; fi
  done
done
python3 - <<PY
import glob, re, statistics, collections
def load(v):
    per = collections.defaultdict(list); tot = []
    for f in sorted(glob.glob('gpurun_out/ab/%s_*.log' % v)):
        for ln in open(f):
            m = re.match(r'\s+(\S.*?)\s+x(\d+)\s+([\d.]+) ms', ln)
            if m: per[m.group(1)].append(float(m.group(3)))
            m = re.search(r'total ([\d.]+) ms', ln)
            if m: tot.append(float(m.group(1)))
    return per, tot
pa, ta = load('A'); pb, tb = load('B')
print('total ms per micro-batch: A %s  B %s' % (ta, tb))
for k in sorted(pa, key=lambda k: -statistics.median(pa[k])):
    a = statistics.median(pa[k]); b = statistics.median(pb.get(k, [float('nan')]))
    print('  %-44s A %.4f  B %.4f  (B/A %.3f)' % (k, a, b, b / a))
print('A = $A, B = $B; medians of $N alternating rounds')
PY
