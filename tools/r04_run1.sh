#!/bin/bash
# round-4 GPU run 1: full GPU suite at HEAD (hygiene changes), per-level stamps of slab_accumulate, same-box baseline bench lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "Warning\|warn" > gpurun_out/r04_gputest_1.txt
tail -5 gpurun_out/r04_gputest_1.txt
LS2FM_LIB=$PWD/tools/ab/lib_stamps.so LS2FM_SERIAL=1 timeout 300 python tools/acc_stamps.py > gpurun_out/r04_acc_stamps_base.txt 2>&1
tail -30 gpurun_out/r04_acc_stamps_base.txt
for cfg in "" "--single-field" "--config C4" "--config C5" "--config C3"; do
  tag=$(echo "base$cfg" | tr -d ' -')
  timeout 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 $cfg 2>/dev/null | tail -1 > gpurun_out/r04_bench_$tag.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r04_bench_$tag.json"))
k=d["roofline"]["all_kernels_avg_us"]
print("$tag", round(d["ms_per_step"],4), d["launch"], {n: round(v,1) for n,v in k.items()})
PY
done
