#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic (bytes per launch).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream,
so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Units are KiB.
    python tools/pmc_traffic.py gpurun_out/pmc3 gpurun_out/pmc4 > profiles/r1_traffic.json"""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                acc[(r['Kernel_Name'], int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
out = []
for (k, grid), c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('FETCH_SIZE', [0]))):
    f = c.get('FETCH_SIZE', [0.0]); w = c.get('WRITE_SIZE', [0.0])
    fetch = 2.0 * 1024.0 * sum(f) / len(f)
    write = 1024.0 * sum(w) / len(w)
    out.append({'kernel': k, 'grid': grid, 'launches': len(f), 'fetch_bytes': round(fetch), 'write_bytes': round(write), 'hbm_bytes': round(fetch + write)})
json.dump(out, sys.stdout, indent=1)
