#!/usr/bin/env python3
"""What the CPU oracle's rate depends on, on the box it is timed on (1.94 frames/s in rounds 3-4, 0.37 in round 5, same
code): `bench.py --cpu-child` in fresh CPU-only processes over thread counts x OpenMP binding policies.  tools/cpu_baseline_probe.py [n_triples]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
styles = sys.argv[2].split(',') if len(sys.argv) > 2 else ['b1']
policies = {'default': {}, 'bind_close_cores': {'OMP_PROC_BIND': 'close', 'OMP_PLACES': 'cores'}, 'bind_spread_cores': {'OMP_PROC_BIND': 'spread', 'OMP_PLACES': 'cores'},
            'close+hugepages': {'OMP_PROC_BIND': 'close', 'OMP_PLACES': 'cores', 'GLIBC_TUNABLES': 'glibc.malloc.hugetlb=1'},
            'default+hugepages': {'GLIBC_TUNABLES': 'glibc.malloc.hugetlb=1'}}
if len(sys.argv) > 3:
    policies = {k: policies[k] for k in sys.argv[3].split(',')}
threads = [int(t) for t in sys.argv[4].split(',')] if len(sys.argv) > 4 else [8, 16, 32, 64, 128]
for style in styles:
    for pol, env_add in policies.items():
        for t in threads:
            env = dict(os.environ, OMP_NUM_THREADS=str(t), MKL_NUM_THREADS=str(t), **env_add)
            env.pop('TTUP_LIB', None)
            t0 = time.time()
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--cpu-child', style, str(n), str(t)], env=env, cwd=ROOT,
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
                out = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
                res = json.loads(out[-1]) if out else {'error': r.stderr[-300:]}
            except subprocess.TimeoutExpired:
                res = {'error': 'timeout'}
            print(style, pol, 'threads', t, 'wall %.1f s' % (time.time() - t0), json.dumps(res), flush=True)
