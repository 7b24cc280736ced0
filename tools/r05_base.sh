#!/bin/bash
# round 5 baseline in one gpurun call: key GPU tests, default bench line, graph-replay timeline, kernel stats  (tag = $1)
TAG=${1:-r05a}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_hip_fused_render.py tests/test_hip_config_shapes.py tests/test_hip_fused_loss.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/${TAG}_tests.txt
cat gpurun_out/${TAG}_tests.txt
python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -c 1500 gpurun_out/${TAG}_bench.json
bash tools/timeline.sh $TAG
LS2FM_SERIAL=1 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -- python bench.py --no-cpu-baseline --launch eager --steps 100 --warmup 10 > /dev/null 2>&1
python tools/prof_top.py gpurun_out/${TAG}_kt 24 > gpurun_out/${TAG}_rocprof_kernel_stats.txt
rm -rf gpurun_out/${TAG}_kt
cat gpurun_out/${TAG}_rocprof_kernel_stats.txt
