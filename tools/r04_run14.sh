#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
F='Warning\|warn(\|amdgpu.ids\|socket.cpp\|Gloo\|^$'
timeout 900 python -m pytest tests/test_hip_fused_render.py tests/test_hip_config_shapes.py tests/test_hip_point_queries.py tests/test_hip_fuzz.py -m gpu -q -x 2>&1 | tail -3
sed -i "s/n in ('wgrad_mlp_sdf','wgrad_mlp_geo')/n in ('slab_accumulate','scatter_fill')/" tools/abenv.sh
echo "== dual"; tools/abenv.sh "" 3 "LS2FM_ACC_PERSISTENT=0" "LS2FM_ACC_PERSISTENT=1" "LS2FM_ACC_DUAL_THREADS=1024" 2>&1 | tee gpurun_out/r04_ab14_persistent_dual.txt
echo "== single"; tools/abenv.sh "--single-field" 2 "LS2FM_ACC_PERSISTENT=0" "LS2FM_ACC_PERSISTENT=1" 2>&1 | tee gpurun_out/r04_ab14_persistent_single.txt
echo "== C3"; tools/abenv.sh "--config C3 --steps 60" 1 "LS2FM_ACC_PERSISTENT=0" "LS2FM_ACC_PERSISTENT=1" "LS2FM_ACC_DUAL_THREADS=1024" 2>&1 | tee gpurun_out/r04_ab14_persistent_c3.txt
echo "== C5"; tools/abenv.sh "--config C5 --steps 60" 1 "LS2FM_ACC_PERSISTENT=0" "LS2FM_ACC_PERSISTENT=1" 2>&1 | tee gpurun_out/r04_ab14_persistent_c5.txt
