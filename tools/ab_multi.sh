#!/bin/bash
# N-way comparison of library builds on ONE box (boxes differ by +-4 %): tools/ops_report.py per build, the builds taking turns.
#   tools/ab_multi.sh <rounds> <lib|default> <lib|default> ...        (lib: a path, or a tag T for upliftingtabletennis_amd/_ablate/libttup_T.so;
#   <lib>+VAR=value runs that build with the environment variable set, e.g. default+TTUP_BB2_GENERIC=1)
# Output: gpurun_out/abm/<tag>_<round>.log and the per-kernel medians of every build side by side.
N=$1; shift
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/abm
rm -f gpurun_out/abm/*.log
for r in $(seq 1 $N); do
  for spec in "$@"; do
    t=${spec%%+*}; ev=""; [ "$t" != "$spec" ] && ev=${spec#*+}
    tag=$(basename "$t" .so); tag=${tag#libttup_}; [ -n "$ev" ] && tag=${tag}+$ev
    lib=$t; [ -f "$lib" ] || lib=upliftingtabletennis_amd/_ablate/libttup_$t.so
    if [ "$t" = default ]; then env -u TTUP_LIB $ev TTUP_REPS=${TTUP_REPS:-10} python3 tools/ops_report.py > "gpurun_out/abm/${tag}_$r.log" 2>&1
    else env $ev TTUP_LIB=$PWD/$lib TTUP_REPS=${TTUP_REPS:-10} python3 tools/ops_report.py > "gpurun_out/abm/${tag}_$r.log" 2>&1; fi
  done
done
python3 - "$@" <<'PY'
import glob, re, statistics, collections, sys, os
tags = []
for spec in sys.argv[1:]:
    t, _, ev = spec.partition('+')
    tag = os.path.basename(t)
    if tag.endswith('.so'): tag = tag[:-3]
    if tag.startswith('libttup_'): tag = tag[8:]
    tags.append(tag + ('+' + ev if ev else ''))
def load(v):
    per = collections.defaultdict(list); tot = []
    for f in sorted(glob.glob('gpurun_out/abm/%s_[0-9]*.log' % glob.escape(v))):
        for ln in open(f):
            m = re.match(r'\s+(\S.*?)\s+x(\d+)\s+([\d.]+) ms', ln)
            if m: per[m.group(1)].append(float(m.group(3)))
            m = re.search(r'total ([\d.]+) ms', ln)
            if m: tot.append(float(m.group(1)))
    return per, tot
data = {t: load(t) for t in tags}
print('builds:', tags)
print('total ms per micro-batch: ' + '   '.join('%s %s' % (t, ['%.4f' % x for x in data[t][1]]) for t in tags))
base = data[tags[0]][0]
for k in sorted(base, key=lambda k: -statistics.median(base[k])):
    row = '  %-40s' % k
    b0 = statistics.median(base[k])
    for t in tags:
        v = data[t][0].get(k)
        row += '  %.4f' % statistics.median(v) if v else '     nan'
        if v and t != tags[0]: row += ' (%.3f)' % (statistics.median(v) / b0)
    print(row)
PY
