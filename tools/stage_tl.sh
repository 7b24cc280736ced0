#!/bin/bash
# kernel timeline of one captured stage step + its timing -> gpurun_out/$1_stage_timeline.txt, $1_stage_step.txt   (tag = $1)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/stl -- python tools/stage_timeline.py run >/dev/null 2>&1
python tools/stage_timeline.py show gpurun_out/stl > gpurun_out/${TAG}_stage_timeline.txt; rm -rf gpurun_out/stl
cat gpurun_out/${TAG}_stage_timeline.txt
python tools/time_stage.py > gpurun_out/${TAG}_stage_step.txt 2>&1; tail -4 gpurun_out/${TAG}_stage_step.txt
