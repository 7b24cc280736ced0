cd $GRAFT_REPO_ROOT
for c in C1 C3 C4 C5; do python bench.py --config $c --no-cpu-baseline > gpurun_out/r03_bench_$c.json 2>/dev/null; done
python bench.py --inference --rays 8192 --no-cpu-baseline > gpurun_out/r03_bench_inference_8192rays.json 2>/dev/null
python bench.py --single-field --no-cpu-baseline > gpurun_out/r03_bench_single_field.json 2>/dev/null
python bench.py --rays 4096 --no-cpu-baseline > gpurun_out/r03_bench_4096rays.json 2>/dev/null
python bench.py --rays 16384 --no-cpu-baseline > gpurun_out/r03_bench_16384rays.json 2>/dev/null
python tools/time_points.py > gpurun_out/r03_point_queries.txt 2>&1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/tl -- python bench.py --no-cpu-baseline --launch graph --steps 50 --warmup 10 >/dev/null 2>&1
python tools/prof_timeline.py gpurun_out/tl 20 > gpurun_out/r03_timeline_graph_replay.txt; rm -rf gpurun_out/tl
for f in gpurun_out/r03_bench_*.json; do python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['launch'])"; done
