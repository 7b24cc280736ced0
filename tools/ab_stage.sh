# A/B of two builds of the library on the captured stage step: tools/ab_stage.sh <alt.so> [runs]
ALT=$1; N=${2:-3}
for i in $(seq $N); do
  for which in cur alt; do
    if [ $which = alt ]; then export LS2FM_LIB=$PWD/$ALT; else unset LS2FM_LIB; fi
    echo "$which $(python tools/time_stage.py 2>/dev/null | grep 'C2' | grep hipGraph | sed 's/.*step *//' | cut -c1-40)"
  done
done
