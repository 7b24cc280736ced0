#!/bin/bash
# a variant build of the library for in-session A/B runs: tools/build_variant.sh <name> "<extra hipcc flags>" <file.hip> [file.hip ...]
#   -> tools/ab/lib_<name>.so (git-ignored, travels with gpurun): the named sources compiled with the extra flags, every other
#   object taken from the regular build (run `make -C level-s2fm_official_amd/csrc` first)
set -e
NAME=$1; FLAGS=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/level-s2fm_official_amd/csrc
OUT=$ROOT/tools/ab; mkdir -p $OUT/obj_$NAME
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics"
OBJS=""
for f in $C/*.hip; do
  b=$(basename $f .hip)
  if [[ " $* " == *" $b.hip "* ]]; then
    PF=""; [ $b = render_fwd ] && PF="-mllvm -amdgpu-sched-strategy=max-memory-clause"     # = the Makefile's FLAGS_<file>
    /opt/rocm/bin/hipcc $BASE $PF $FLAGS -c $f -o $OUT/obj_$NAME/$b.o
    OBJS="$OBJS $OUT/obj_$NAME/$b.o"
  else
    OBJS="$OBJS $C/_build/$b.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$NAME.so $OBJS
echo "built $OUT/lib_$NAME.so"
