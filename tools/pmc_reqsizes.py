#!/usr/bin/env python3
"""Read-side HBM traffic per kernel from the L2's request-size counters (separate --pmc pass):
    rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B -d out -- python3 tools/prof_cnn.py
    python tools/pmc_reqsizes.py out [fetch_pass_dir]
bytes = 32 * n32 + 64 * n64 + 128 * n128 (requests of no size class, if any, are listed).  Beside it: FETCH_SIZE x 2 of the other pass --
the guide's gfx950 correction holds for wide coalesced streams only, this checks it on each kernel's own access pattern."""
import collections, csv, glob, json, sys
def load(d, names):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] in names:
                acc[(r['Kernel_Name'], int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc
N = ('TCC_EA0_RDREQ', 'TCC_EA0_RDREQ_32B', 'TCC_EA0_RDREQ_64B', 'TCC_EA0_RDREQ_128B', 'TCC_EA0_RDREQ_DRAM_32B', 'TCC_EA0_RDREQ_DRAM', 'TCC_HIT', 'TCC_MISS')
a = load(sys.argv[1], N)
fs = load(sys.argv[2], ('FETCH_SIZE',)) if len(sys.argv) > 2 else {}
out = []
for key, c in a.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    row = {'kernel': key[0], 'grid': key[1], 'launches': len(next(iter(c.values())))}
    row.update({k.replace('TCC_EA0_', '').lower(): round(v) for k, v in m.items()})
    if 'TCC_EA0_RDREQ_128B' in m:
        row['read_bytes_by_size'] = round(32 * m.get('TCC_EA0_RDREQ_32B', 0) + 64 * m.get('TCC_EA0_RDREQ_64B', 0) + 128 * m['TCC_EA0_RDREQ_128B'])
        row['requests_without_size_class'] = round(m.get('TCC_EA0_RDREQ', 0) - m.get('TCC_EA0_RDREQ_32B', 0) - m.get('TCC_EA0_RDREQ_64B', 0) - m['TCC_EA0_RDREQ_128B'])
    if 'TCC_EA0_RDREQ_DRAM_32B' in m: row['dram_read_bytes'] = round(32 * m['TCC_EA0_RDREQ_DRAM_32B'])
    if key in fs: row['fetch_size_x2_bytes'] = round(2048 * sum(fs[key]['FETCH_SIZE']) / len(fs[key]['FETCH_SIZE']))
    out.append(row)
out.sort(key=lambda r: -r.get('read_bytes_by_size', r.get('dram_read_bytes', 0)))
json.dump(out, sys.stdout, indent=1)
