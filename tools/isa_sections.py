#!/usr/bin/env python3
"""Static instruction mix of one kernel of a `hipcc -S` listing PER BARRIER-DELIMITED SECTION (tools/isa_sections.py conv.s <kernel name substring>):
for the fully unrolled tile loops of the persistent kernels this is what a wave issues per phase.  VALU-issue cycles ~ 4 x valu + 8 x mfma per wave."""
import collections, re, sys
lines = [l.rstrip('\n') for l in open(sys.argv[1])]
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and l.split(';')[0].strip().endswith(':'))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
def cls(op):
    for p, c in (('v_mfma', 'mfma'), ('ds_read', 'lds_r'), ('ds_load', 'lds_r'), ('ds_', 'lds_w'), ('global_load', 'vm_ld'), ('buffer_load', 'vm_ld'), ('scratch_load', 'vm_ld'),
                 ('global_store', 'vm_st'), ('buffer_store', 'vm_st'), ('scratch_store', 'vm_st'), ('s_waitcnt', 'wait'), ('s_barrier', 'barrier'), ('s_cbranch', 'br'), ('s_branch', 'br'),
                 ('s_nop', 'nop'), ('s_', 'salu'), ('v_', 'valu')):
        if op.startswith(p):
            return c
    return 'other'
sec, secs = collections.Counter(), []
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith((';', '.', '//')) or t.endswith(':'):
        continue
    c = cls(t.split()[0])
    if c == 'barrier':
        secs.append(sec); sec = collections.Counter()
    else:
        sec[c] += 1
secs.append(sec)
print(lines[start].split(':')[0])
for i, s in enumerate(secs):
    print('  section %d: ' % i + '  '.join('%s %d' % (k, s[k]) for k in ('mfma', 'valu', 'salu', 'lds_r', 'lds_w', 'vm_ld', 'vm_st', 'wait', 'br', 'nop') if s[k]) +
          '   | vector issue ~ %d cycles' % (4 * s['valu'] + 8 * s['mfma']))
