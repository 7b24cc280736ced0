cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS="bench.py --no-cpu-baseline --launch eager --steps 20 --warmup 5"
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-20)
  LS2FM_SERIAL=1 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/d_$tag -- python $ARGS > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/d_$tag 2>&1 | grep -E "shade_bwd|slab_acc|index_fill|shade_fwd|ray_encode" 
  rm -rf gpurun_out/d_$tag
done
