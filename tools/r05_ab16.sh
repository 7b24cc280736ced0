#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abenv.sh "--launch graph" 2 "" "LS2FM_PROBE_NO_SIDE=1"
bash tools/abenv.sh "--launch eager" 2 "" "LS2FM_PROBE_NO_SIDE=1"
bash tools/abenv.sh "--config C1" 2 "" "LS2FM_FUSED_WGRAD=0"
LS2FM_PROBE_NO_SIDE=1 rocprofv3 --kernel-trace -d gpurun_out/tl -- python bench.py --no-cpu-baseline --launch graph --steps 50 --warmup 10 >/dev/null 2>&1
python tools/prof_timeline.py gpurun_out/tl 20; rm -rf gpurun_out/tl
