#!/bin/bash
# A/B of environment settings in one session: tools/abenv.sh "<bench args>" <runs> "" "VAR=1" "VAR=2 OTHER=x" ...
ARGS=$1; N=$2; shift; shift
for i in $(seq $N); do
  for setting in "$@"; do
    env $setting python bench.py --no-cpu-baseline --steps 300 --warmup 30 $ARGS 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_avg_us']
print('[$setting]'.ljust(28), round(d['ms_per_step'],4), d['launch'][:5], {n:round(v,1) for n,v in k.items() if n in ('slab_accumulate','scatter_fill','wgrad_dec','reduce_finalize')})"
  done
done
