#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_fused_render.py tests/test_hip_config_shapes.py tests/test_hip_fused_loss.py tests/test_hip_stage.py -m gpu -x -q 2>&1 | tail -4
bash tools/abenv.sh "" 3 "LS2FM_SIDE_IN_FILL=0" "LS2FM_SIDE_IN_FILL=1"
bash tools/abenv.sh "--config C1" 1 "LS2FM_SIDE_IN_FILL=0" "LS2FM_SIDE_IN_FILL=1" "LS2FM_FUSED_WGRAD=0"
bash tools/abenv.sh "--single-field" 1 "LS2FM_SIDE_IN_FILL=0" "LS2FM_SIDE_IN_FILL=1"
bash tools/timeline.sh r05c | tail -9
