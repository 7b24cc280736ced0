#!/bin/bash
# kernel timelines of the benchmark step: graph replay and eager launches, side chain enqueued first / last
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for mode in graph eager; do for sf in 1 0; do
  LS2FM_SIDE_FIRST=$sf rocprofv3 --kernel-trace -d gpurun_out/tl -- python bench.py --no-cpu-baseline --launch $mode --steps 50 --warmup 10 >/dev/null 2>&1
  echo "== launch $mode, LS2FM_SIDE_FIRST=$sf"
  python tools/prof_timeline.py gpurun_out/tl 20; rm -rf gpurun_out/tl
done; done
