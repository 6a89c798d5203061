cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
for v in 0 1; do
  O=$R/gpurun_out/s2xcd$v; rm -rf $O; mkdir -p $O
  TTUP_S2_XCD=$v TTUP_PROF_REPS=20 rocprofv3 --output-format csv --kernel-trace --stats -d $O/ktc -o kt -- python3 tools/prof_cnn.py > $O/ktc.log 2>&1
  f=$(find $O/ktc -name '*kernel_stats.csv' | head -1); cp $f $O/ktc.csv; rm -rf $O/ktc
  echo "== TTUP_S2_XCD=$v"; python3 - <<PY
import csv
for r in csv.DictReader(open('$O/ktc.csv')):
    if ', 2, 4, ' in r['Name'] or 'bneck' in r['Name'] or 's2_pair' in r['Name']: print('%-75s calls %4s avg_us %8.1f'%(r['Name'][:75], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
