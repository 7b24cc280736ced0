#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/abenv.sh "--single-field" 2 "LS2FM_SIDE_IN_FILL=0" ""
bash tools/abenv.sh "--config C1" 1 ""
bash tools/abenv.sh "--config C5" 1 "LS2FM_SIDE_IN_FILL=0" ""
bash tools/abenv.sh "--config C3" 1 "LS2FM_SIDE_IN_FILL=0" ""
bash tools/abenv.sh "--launch eager" 2 "LS2FM_SIDE_IN_FILL=0" ""
timeout 1200 python -m pytest tests/test_hip_fused_loss.py tests/test_hip_stage.py tests/test_hip_dist_two_rank.py tests/test_hip_graph_capture.py tests/test_hip_checkpoint.py -m gpu -x -q 2>&1 | tail -3
