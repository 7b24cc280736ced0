# A/B of two builds of the library in one session with extra bench arguments: tools/ab2.sh <alt.so> <runs> [bench args]
ALT=$1; N=$2; shift; shift
for i in $(seq $N); do
  for which in cur alt; do
    if [ $which = alt ]; then export LS2FM_LIB=$PWD/$ALT; else unset LS2FM_LIB; fi
    python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_avg_us']; print('$which', round(d['ms_per_step'],4), {n:k[n] for n in k if n in ('slab_accumulate','scatter_fill')})"
  done
done
