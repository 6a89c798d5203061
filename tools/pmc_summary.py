#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean of each counter over dispatches."""
import collections
import csv
import glob
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            rows[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in sorted(rows.items(), key=lambda kv: -sum(kv[1].get('SQ_WAVE_CYCLES', [0]))):
    print(k[:120])
    print('   ' + '  '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(cs.items())) + '  (n=%d)' % len(next(iter(cs.values()))))
