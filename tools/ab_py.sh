# A/B of two builds with the SAME python side: tools/ab_py.sh <alt.so> [runs]   (prints forward-side kernels too)
ALT=$1; N=${2:-3}
for i in $(seq $N); do
  for which in cur alt; do
    if [ $which = alt ]; then export LS2FM_LIB=$PWD/$ALT; else unset LS2FM_LIB; fi
    python bench.py --no-cpu-baseline --launch eager --steps 300 --warmup 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_avg_us']; print('$which', round(d['ms_per_step'],4), {n:k[n] for n in k if 'shade' in n or 'encode' in n or 'prep' in n})"
  done
done
