#!/bin/bash
# Build variants of libttup.so with -DTTUP_<flag> on ONE source (conv.hip unless SRC=<name> is set), e.g. ABLATE_S1 or TIMING for
# phase-by-phase timing; FLAG_A,FLAG_B gives one library with both.
#   tools/build_ablate.sh ABLATE_S1 TIMING,ABL_2A_NOMFMA  ->  upliftingtabletennis_amd/_ablate/libttup_ABLATE_S1.so ...   (select with TTUP_LIB)
#   SRC=conv_x3 tools/build_ablate.sh NO_FRAG_PIPELINE    ->  _ablate/libttup_conv_x3_NO_FRAG_PIPELINE.so
#   BASE=1 tools/build_ablate.sh X                        ->  also links the git HEAD version of the source as libttup_<src>_HEAD.so
set -e
cd "$(dirname "$0")/../upliftingtabletennis_amd"
mkdir -p _ablate
SRC=${SRC:-conv}
OBJS="api conv conv_f32 conv_x3 refine wasb_net certify uplift trajgen odefit calib peaks"
for f in "$@"; do
  defs=""; for d in ${f//,/ }; do defs="$defs -DTTUP_$d"; done          # FLAG_A,FLAG_B -> -DTTUP_FLAG_A -DTTUP_FLAG_B (one library, named after the list)
  tag=$f; [ "$SRC" != conv ] && tag=${SRC}_$f
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-fast-math $defs -c csrc/$SRC.hip -o _ablate/${SRC}_$f.o
  link=""; for o in $OBJS; do if [ $o = $SRC ]; then link="$link _ablate/${SRC}_$f.o"; else link="$link csrc/$o.o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _ablate/libttup_$tag.so $link
  rm _ablate/${SRC}_$f.o
done
ls -la _ablate
