#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_fused_render.py tests/test_hip_config_shapes.py tests/test_hip_point_queries.py tests/test_hip_stage.py tests/test_hip_camera_rays.py -m gpu -x -q 2>&1 | tail -4
bash tools/abenv.sh "" 3 ""
bash tools/abenv.sh "--single-field" 1 ""
bash tools/abenv.sh "--config C5" 1 ""
bash tools/timeline.sh r05b | tail -12
