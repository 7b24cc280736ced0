cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS="bench.py --no-cpu-baseline --launch eager --steps 10 --warmup 3"
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-20)
  LS2FM_SERIAL=1 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/d_$tag -- python $ARGS > gpurun_out/pmc_$tag.log 2>&1
  python tools/pmc_summary.py gpurun_out/d_$tag 2>&1 | grep -E "${KFILTER:-shade_bwd}"
  rm -rf gpurun_out/d_$tag
done
