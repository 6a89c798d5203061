#!/bin/bash
# Build variants of libttup.so with one -DTTUP_<flag> each, e.g. ABLATE_S1 or BB_UNROLL (conv.hip only) for phase-by-phase timing.
#   tools/build_ablate.sh ABLATE_S1 ABLATE_S2  ->  upliftingtabletennis_amd/_ablate/libttup_ABLATE_S1.so ...   (select with TTUP_LIB)
set -e
cd "$(dirname "$0")/../upliftingtabletennis_amd"
mkdir -p _ablate
for f in "$@"; do
  defs=""; for d in ${f//,/ }; do defs="$defs -DTTUP_$d"; done          # FLAG_A,FLAG_B -> -DTTUP_FLAG_A -DTTUP_FLAG_B (one library, named after the list)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-fast-math $defs -c csrc/conv.hip -o _ablate/conv_$f.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ablate/libttup_$f.so csrc/api.o _ablate/conv_$f.o csrc/conv_f32.o csrc/conv_x3.o csrc/refine.o csrc/wasb_net.o csrc/certify.o csrc/uplift.o csrc/trajgen.o csrc/odefit.o csrc/calib.o csrc/peaks.o
  rm _ablate/conv_$f.o
done
ls -la _ablate
