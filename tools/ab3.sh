# in-session comparison of several builds, eager and graph: tools/ab3.sh lib1.so lib2.so ...
for i in 1 2 3; do
  for lib in "$@"; do
    for m in eager graph; do
      LS2FM_LIB=$PWD/$lib python bench.py --no-cpu-baseline --launch $m --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', '$m', round(d['ms_per_step'],4))"
    done
  done
done
