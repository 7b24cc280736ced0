#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['launch'], d['launch_probe'], d['roofline']['all_kernels_avg_us'])"; done
python bench.py --no-cpu-baseline --launch eager 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('eager', round(d['ms_per_step'],4), d['launch'])"
