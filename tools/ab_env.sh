# A/B of an environment setting in one session: tools/ab_env.sh "VAR=value [VAR2=value2]" [runs] [extra bench args]
SET=$1; N=${2:-3}; shift; shift
for i in $(seq $N); do
  for which in base set; do
    if [ $which = set ]; then PRE="env $SET"; else PRE=""; fi
    $PRE python bench.py --no-cpu-baseline --steps 300 --warmup 30 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which', round(d['ms_per_step'],4), d['launch'], d.get('launch_probe'))"
  done
done
