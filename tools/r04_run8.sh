#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
echo "== dual"; tools/abn.sh "" 3 cur tools/ab/lib_head.so 2>&1 | tee gpurun_out/r04_ab8_wgil_dual.txt
echo "== single"; tools/abn.sh "--single-field" 2 cur tools/ab/lib_head.so 2>&1 | tee gpurun_out/r04_ab8_wgil_single.txt
timeout 600 python -m pytest tests/test_hip_fused_render.py -m gpu -q -x 2>&1 | tail -2
