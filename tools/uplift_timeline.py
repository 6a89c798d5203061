#!/usr/bin/env python3
"""Timeline of one small-batch uplift forward from a rocprofv3 kernel trace: per-kernel duration and the gap to the previous kernel.
    cd /tmp; TTUP_UPLIFT_B=1 TTUP_UPLIFT_T=50 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/bench_uplift.py
    python tools/uplift_timeline.py $OUT"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# the last forward: walk back from the end to the previous prepare_kernel
last = max(i for i, r in enumerate(rows) if 'prepare_kernel' in r[2])
seg = rows[last:]
t0 = seg[0][0]
busy = 0
for i, (s, e, n) in enumerate(seg):
    gap = s - seg[i - 1][1] if i else 0
    busy += e - s
    print('%8.1f us  +%6.1f gap  %6.1f us  %s' % ((s - t0) / 1e3, gap / 1e3, (e - s) / 1e3, n.split('(')[0][-60:]))
print('span %.1f us, busy %.1f us, %d kernels' % ((seg[-1][1] - t0) / 1e3, busy / 1e3, len(seg)))
