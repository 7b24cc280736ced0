#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_hip_fused_render.py tests/test_hip_config_shapes.py tests/test_hip_point_queries.py -m gpu -x -q 2>&1 | tail -3
bash tools/abenv.sh "" 2 ""
bash tools/abenv.sh "--single-field" 2 ""
bash tools/abenv.sh "--config C5" 1 ""
bash tools/abenv.sh "--config C3" 1 ""
bash tools/abenv.sh "--config C4" 1 ""
