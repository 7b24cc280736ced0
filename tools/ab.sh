#!/bin/bash
# In-session A/B/C... of library builds and / or environment settings (boxes differ by 2 - 3 %: only lines of ONE gpurun call compare).
#   tools/ab.sh "<bench args>" <runs> <variant> [<variant> ...]
# a variant is "cur" (the in-tree build), a path to another build (tools/build_variant.sh -> tools/ab/lib_<name>.so), or a quoted
# list of environment settings, optionally with a library first:   cur   tools/ab/lib_x.so   "LS2FM_EXPLICIT_LEVELS=0"   "tools/ab/lib_x.so LS2FM_ACC_HOLD=0"
# prints ms/step, the launch form and bench.py's per-kernel spans (us) of every run
ARGS=$1; N=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd $ROOT
for i in $(seq $N); do
  for v in "$@"; do
    LIB=""; ENVS=""
    for w in $v; do case $w in cur) ;; *.so) LIB=$ROOT/$w ;; *) ENVS="$ENVS $w" ;; esac; done
    env ${LIB:+LS2FM_LIB=$LIB} $ENVS python bench.py --no-cpu-baseline --steps 300 --warmup 30 $ARGS 2>/dev/null | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_avg_us']
    print('[$v]'.ljust(44), round(d['ms_per_step'],4), d['launch'][:5], {n:round(x,1) for n,x in k.items()})
except Exception as e: print('[$v]', 'FAILED', e)"
  done
done
