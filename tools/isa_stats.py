#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -S listing (tools/isa_stats.py conv.s <kernel name substring> [--top N]).
The chain kernels are fully unrolled, so the static mix is close to what a wave executes per tile."""
import collections
import re
import sys


def body(lines, key):
    start = None
    for i, ln in enumerate(lines):
        head = ln.split(';')[0].strip()
        if head.endswith(':') and key in head and not head.startswith('.'):
            start = i
            break
    if start is None:
        raise SystemExit('kernel %r not found' % key)
    out = []
    for ln in lines[start + 1:]:
        s = ln.strip()
        if s.startswith('s_endpgm'):
            break
        out.append(s)
    return lines[start], out


def classify(op):
    if op.startswith('v_mfma'):
        return 'mfma'
    if op.startswith('ds_read') or op.startswith('ds_load'):
        return 'lds_read'
    if op.startswith('ds_'):
        return 'lds_write'
    if op.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')):
        return 'vmem_load'
    if op.startswith(('global_store', 'buffer_store', 'flat_store', 'scratch_store', 'global_atomic')):
        return 'vmem_store'
    if op.startswith('s_waitcnt'):
        return 'waitcnt'
    if op.startswith('s_barrier'):
        return 'barrier'
    if op.startswith(('s_cbranch', 's_branch')):
        return 'branch'
    if op.startswith('s_nop'):
        return 's_nop'
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
        return 'smem'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('v_'):
        return 'valu'
    return 'other'


def main():
    path, key = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 25
    lines = [ln.rstrip('\n') for ln in open(path)]
    name, ins = body(lines, key)
    cls = collections.Counter()
    ops = collections.Counter()
    for s in ins:
        if not s or s.startswith((';', '.', '//')) or s.endswith(':'):
            continue
        op = re.split(r'\s+', s)[0]
        c = classify(op)
        cls[c] += 1
        if c in ('valu', 'salu'):
            ops[op] += 1
    print(name)
    print('  ' + '  '.join('%s %d' % kv for kv in sorted(cls.items(), key=lambda kv: -kv[1])))
    print('  top vector/scalar ALU opcodes: ' + ', '.join('%s %d' % kv for kv in ops.most_common(top)))


if __name__ == '__main__':
    main()
