#!/bin/bash
# kernel timeline of one graph replay of the benchmark step -> gpurun_out/$1_timeline_graph_replay.txt (tag = $1)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/tl -- python bench.py --no-cpu-baseline --launch graph --steps 50 --warmup 10 >/dev/null 2>&1
python tools/prof_timeline.py gpurun_out/tl 20 > gpurun_out/${TAG}_timeline_graph_replay.txt; rm -rf gpurun_out/tl
cat gpurun_out/${TAG}_timeline_graph_replay.txt
