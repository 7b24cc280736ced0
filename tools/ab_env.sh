# A/B of one environment variable in one session: tools/ab_env.sh VAR valA valB [runs]
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for val in $A $B; do
    env $VAR=$val python bench.py --no-cpu-baseline --launch eager --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_avg_us']; print('$VAR=$val', round(d['ms_per_step'],4), {n:k[n] for n in k if 'encode' in n or n in ('shade_fwd','prep_weights')})"
  done
done
