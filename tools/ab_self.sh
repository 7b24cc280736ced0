# repeated bench runs of the current build: tools/ab_self.sh [runs]
N=${1:-3}
for i in $(seq $N); do
  python bench.py --no-cpu-baseline --launch eager --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_avg_us']; print('cur', round(d['ms_per_step'],4), {n:k[n] for n in k if 'scatter' in n or 'slab' in n or n in ('shade_bwd','bin_build')})"
done
