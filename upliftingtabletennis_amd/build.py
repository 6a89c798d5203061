"""Build libttup.so (HIP, gfx950) in-tree with hipcc.  No CPU fallback is built: the product path is the
HIP library only.

    python -m upliftingtabletennis_amd.build [--force]
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libttup.so')
SOURCES = ['api.hip', 'conv.hip', 'conv_f32.hip', 'refine.hip', 'wasb_net.hip', 'certify.hip', 'uplift.hip', 'trajgen.hip', 'odefit.hip', 'calib.hip', 'peaks.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-fno-fast-math']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every .hip source to an object and link the shared library.  Returns the library path."""
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + \
              [os.path.join(os.path.dirname(HERE), 'include', 'ttup.h')]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stdout))
        return r.stdout
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for out in ex.map(run, jobs):
            if verbose and out.strip():
                print(out)
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
