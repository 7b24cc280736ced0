#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abenv.sh "" 2 "LS2FM_SIDE_IN_FILL=0" "" 
bash tools/abenv.sh "" 1 "LS2FM_SIDE_PROBE=1" "LS2FM_SIDE_PROBE=2" "LS2FM_SIDE_PROBE=6"
timeout 900 python -m pytest tests/test_hip_fused_render.py tests/test_hip_config_shapes.py -m gpu -x -q 2>&1 | tail -3
