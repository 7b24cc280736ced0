#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
sed -i "s/n in ('slab_accumulate','scatter_fill','wgrad_dec','reduce_finalize')/n in ('wgrad_mlp_sdf','wgrad_mlp_geo')/" tools/abenv.sh
echo "== dual"; tools/abenv.sh "" 2 "A=0" "LS2FM_EXP_FILL_AFTER=1" "LS2FM_EXP_FILL_AFTER=2" "LS2FM_EXP_WGRAD_BLOCKS=512" "LS2FM_EXP_WGRAD_BLOCKS=512 LS2FM_EXP_FILL_AFTER=2" "LS2FM_EXP_WGRAD_BLOCKS=512 LS2FM_EXP_FILL_AFTER=1" 2>&1 | tee gpurun_out/r04_ab7_sched_dual.txt
echo "== single"; tools/abenv.sh "--single-field" 2 "A=0" "LS2FM_EXP_FILL_AFTER=2" "LS2FM_EXP_WGRAD_BLOCKS=512" "LS2FM_EXP_WGRAD_BLOCKS=512 LS2FM_EXP_FILL_AFTER=2" 2>&1 | tee gpurun_out/r04_ab7_sched_single.txt
