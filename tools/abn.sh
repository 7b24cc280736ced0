#!/bin/bash
# A/B/C... of library builds in one session: tools/abn.sh "<bench args>" <runs> cur tools/ab/lib_x.so tools/ab/lib_y.so ...
ARGS=$1; N=$2; shift; shift
for i in $(seq $N); do
  for lib in "$@"; do
    if [ $lib = cur ]; then unset LS2FM_LIB; else export LS2FM_LIB=$PWD/$lib; fi
    python bench.py --no-cpu-baseline --steps 300 --warmup 30 $ARGS 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_avg_us']
print('$(basename $lib .so)'.ljust(16), round(d['ms_per_step'],4), d['launch'][:5], {n:round(v,1) for n,v in k.items()})"
  done
done
