"""Build libttup.so (HIP, gfx950) in-tree with hipcc.  No CPU fallback is built: the product path is the
HIP library only.

    python -m upliftingtabletennis_amd.build [--force]

The library carries the hash of the sources it was built from (`ttup_build_id()`): `source_id()` below hashes the SOURCES under csrc/,
csrc/*.h, include/ttup.h and the compiler flags; it is compiled into api.o, `_lib.load()` refuses a library whose id differs
from the tree's, and an object is rebuilt when the hash of ITS inputs differs from the one recorded for it (csrc/.stamps.json)
-- modification times are not consulted.
"""
import concurrent.futures
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libttup.so')
STAMPS = os.path.join(CSRC, '.stamps.json')
SOURCES = ['api.hip', 'conv.hip', 'conv_f32.hip', 'conv_x3.hip', 'refine.hip', 'wasb_net.hip', 'certify.hip', 'uplift.hip', 'trajgen.hip', 'odefit.hip', 'calib.hip', 'peaks.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function',
         '-fno-fast-math']


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + \
        [os.path.join(os.path.dirname(HERE), 'include', 'ttup.h')]


def _digest(paths, extra=''):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b'\0')
        with open(p, 'rb') as f:
            h.update(hashlib.sha256(f.read()).digest())
    return h.hexdigest()


def source_id():
    """16 hex digits identifying the library's inputs: every .hip / .h under csrc/, include/ttup.h, the flags."""
    srcs = sorted(os.path.join(CSRC, f) for f in SOURCES)          # exactly what build() compiles: a stray .hip in csrc/ does not change the id
    return _digest(srcs + _headers(), ' '.join(FLAGS))[:16]


def build(force=False, verbose=False):
    """Compile every .hip source whose inputs changed and link the shared library.  Returns the library path."""
    headers = _headers()
    sid = source_id()
    try:
        with open(STAMPS) as f:
            stamps = json.load(f)
    except (OSError, ValueError):
        stamps = {}
    objs, jobs, new_stamps = [], [], {}
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(CSRC, src.replace('.hip', '.o'))
        objs.append(o)
        flags = FLAGS + (['-DTTUP_BUILD_ID="%s"' % sid] if src == 'api.hip' else [])
        stamp = _digest([s] + headers, ' '.join(flags))
        new_stamps[src] = stamp
        if force or not os.path.exists(o) or stamps.get(src) != stamp:
            jobs.append([HIPCC] + flags + ['-c', s, '-o', o])

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
    if force or jobs or not os.path.exists(LIB) or stamps.get('lib') != sid:
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    new_stamps['lib'] = sid
    if new_stamps != stamps:
        with open(STAMPS, 'w') as f:
            json.dump(new_stamps, f, indent=0, sort_keys=True)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True), source_id())
