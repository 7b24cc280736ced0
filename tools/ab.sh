# A/B of two builds of the library in one session: tools/ab.sh <alt.so> [runs]
ALT=$1; N=${2:-3}
for i in $(seq $N); do
  for which in cur alt; do
    if [ $which = alt ]; then export LS2FM_LIB=$PWD/$ALT; else unset LS2FM_LIB; fi
    python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_avg_us']; print('$which', round(d['ms_per_step'],4), {n:k[n] for n in k if n in ('slab_accumulate','scatter_fill','wgrad_mlp_sdf','wgrad_mlp_geo','reduce_finalize','wgrad_dec')})"
  done
done
