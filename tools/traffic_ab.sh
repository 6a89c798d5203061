cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for tag in new prev; do
  if [ $tag = prev ]; then export TTUP_LIB=$R/upliftingtabletennis_amd/_ablate/libttup_PREV.so; fi
  rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d gpurun_out/tab_$tag/f -- python3 tools/prof_cnn.py > /dev/null 2>&1
  rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d gpurun_out/tab_$tag/w -- python3 tools/prof_cnn.py > /dev/null 2>&1
  python3 tools/pmc_traffic.py gpurun_out/tab_$tag/f gpurun_out/tab_$tag/w > gpurun_out/tab_$tag.json
done
