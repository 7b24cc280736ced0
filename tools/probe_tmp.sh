run() { # name lib env
  if [ "$2" = cur ]; then unset LS2FM_LIB; else export LS2FM_LIB=$PWD/$2; fi
  env $3 python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['roofline']['all_kernels_avg_us']
print('$1'.ljust(24), round(d['ms_per_step'],4), {n:round(v,1) for n,v in k.items() if n in ('shade_bwd','wgrad_dec','reduce_finalize','scatter_fill')})"
}
python -m pytest tests/test_hip_fused_render.py tests/test_hip_config_shapes.py tests/test_hip_fused_loss.py tests/test_hip_stage.py -m gpu -x -q 2>&1 | tail -3
run fused cur A=1
run unfused cur LS2FM_FUSED_WGRAD=0
run fused cur A=1
run unfused cur LS2FM_FUSED_WGRAD=0
