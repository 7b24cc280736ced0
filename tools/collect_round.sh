#!/bin/bash
# one gpurun call: everything profiles/ holds for a round  ->  gpurun_out/<tag>_*   (tag = $1, default r06)
#   kernel-trace stats, PMC passes (FETCH_SIZE, WRITE_SIZE, L2 hit / miss, MFMA busy), the default bench line (with the CPU leg),
#   secondary bench lines, graph-replay timelines, stage / loop timings and timelines, point queries
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ARGS="bench.py --no-cpu-baseline --launch eager"
LS2FM_SERIAL=1 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -- python $ARGS --steps 100 --warmup 10 > gpurun_out/${TAG}_kt_bench.json 2>/dev/null
python tools/prof_top.py gpurun_out/${TAG}_kt 24 > gpurun_out/${TAG}_rocprof_kernel_stats.txt
tail -1 gpurun_out/${TAG}_kt_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py spans of the same run (us):', d['roofline']['all_kernels_avg_us'])" >> gpurun_out/${TAG}_rocprof_kernel_stats.txt
rm -rf gpurun_out/${TAG}_kt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-12)
  LS2FM_SERIAL=1 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/${TAG}_pmc_$tag -- python $ARGS --steps 20 --warmup 5 > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/${TAG}_pmc_$tag > gpurun_out/${TAG}_pmc_$tag.txt 2>&1
  rm -rf gpurun_out/${TAG}_pmc_$tag
done
mv gpurun_out/${TAG}_pmc_FETCH_SIZE.txt gpurun_out/${TAG}_pmc_fetch.txt; mv gpurun_out/${TAG}_pmc_WRITE_SIZE.txt gpurun_out/${TAG}_pmc_write.txt
mv gpurun_out/${TAG}_pmc_TCC_HIT_sum_.txt gpurun_out/${TAG}_pmc_l2.txt; mv gpurun_out/${TAG}_pmc_SQ_INSTS_VAL.txt gpurun_out/${TAG}_pmc_mfma.txt
mkdir -p profiles; python tools/pmc_to_json.py ${TAG} ${LS2FM_COMMIT:-unknown} > /dev/null 2>&1; cp profiles/${TAG}_pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
for c in C1 C3 C4 C5; do python bench.py --config $c --no-cpu-baseline > gpurun_out/${TAG}_bench_$c.json 2>/dev/null; done
python bench.py --inference --rays 8192 --no-cpu-baseline > gpurun_out/${TAG}_bench_inference_8192rays.json 2>/dev/null
python bench.py --single-field --no-cpu-baseline > gpurun_out/${TAG}_bench_single_field.json 2>/dev/null
python bench.py --rays 4096 --no-cpu-baseline > gpurun_out/${TAG}_bench_4096rays.json 2>/dev/null
python bench.py --rays 16384 --no-cpu-baseline > gpurun_out/${TAG}_bench_16384rays.json 2>/dev/null
python bench.py --with-update --no-cpu-baseline > gpurun_out/${TAG}_bench_with_update.json 2>/dev/null
# the N > 1 code path on a one-rank RCCL group (collectives are identities, everything else is real): the default form (the
# metric's step: fwd + bwd + all-reduce, launched like N = 1) eager and captured, and the opt-in sharded form
LS2FM_DIST_SINGLE=1 python bench.py --force-dist --no-cpu-baseline > gpurun_out/${TAG}_bench_1rank_rccl_allreduce_auto.json 2>/dev/null
LS2FM_DIST_SINGLE=1 python bench.py --force-dist --no-cpu-baseline --launch eager > gpurun_out/${TAG}_bench_1rank_rccl_allreduce_eager.json 2>/dev/null
LS2FM_DIST_SINGLE=1 python bench.py --force-dist --no-cpu-baseline --launch graph --capture-overlap > gpurun_out/${TAG}_bench_1rank_rccl_allreduce_graph_level_groups.json 2>/dev/null
rocprofv3 --kernel-trace -d gpurun_out/tl -- env LS2FM_DIST_SINGLE=1 python bench.py --force-dist --no-cpu-baseline --launch graph --capture-overlap --steps 50 --warmup 10 >/dev/null 2>&1
python tools/prof_timeline.py gpurun_out/tl 30 > gpurun_out/${TAG}_timeline_graph_replay_1rank_rccl_level_groups.txt; rm -rf gpurun_out/tl
LS2FM_DIST_SINGLE=1 python bench.py --force-dist --no-cpu-baseline --launch graph > gpurun_out/${TAG}_bench_1rank_rccl_allreduce_graph.json 2>/dev/null
LS2FM_DIST_SINGLE=1 python bench.py --force-dist --no-cpu-baseline --shard > gpurun_out/${TAG}_bench_1rank_rccl_shard_monolithic.json 2>/dev/null
LS2FM_DIST_SINGLE=1 python bench.py --force-dist --no-cpu-baseline --shard --launch graph > gpurun_out/${TAG}_bench_1rank_rccl_shard_monolithic_graph.json 2>/dev/null
python tools/time_points.py > gpurun_out/${TAG}_point_queries.txt 2>&1
[ -f tools/ab/lib_stamps.so ] && LS2FM_LIB=$PWD/tools/ab/lib_stamps.so timeout 300 python tools/acc_stamps_p.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_acc_stamps.txt
python tools/time_stage.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_stage_step.txt
python tools/time_loops.py 2>&1 | grep -v amdgpu > gpurun_out/${TAG}_loops_step.txt
rocprofv3 --kernel-trace -d gpurun_out/tl -- python bench.py --no-cpu-baseline --launch graph --steps 50 --warmup 10 >/dev/null 2>&1
python tools/prof_timeline.py gpurun_out/tl 20 > gpurun_out/${TAG}_timeline_graph_replay.txt; rm -rf gpurun_out/tl
rocprofv3 --kernel-trace -d gpurun_out/tl -- python bench.py --no-cpu-baseline --launch graph --steps 50 --warmup 10 --single-field >/dev/null 2>&1
python tools/prof_timeline.py gpurun_out/tl 20 > gpurun_out/${TAG}_timeline_graph_replay_single_field.txt; rm -rf gpurun_out/tl
rocprofv3 --kernel-trace -d gpurun_out/stl -- python tools/stage_timeline.py run >/dev/null 2>&1
python tools/stage_timeline.py show gpurun_out/stl > gpurun_out/${TAG}_stage_timeline.txt; rm -rf gpurun_out/stl
rocprofv3 --kernel-trace -d gpurun_out/ltl -- python tools/loop_timeline.py run BA >/dev/null 2>&1
python tools/loop_timeline.py show gpurun_out/ltl > gpurun_out/${TAG}_ba_loop_timeline.txt; rm -rf gpurun_out/ltl
for f in gpurun_out/${TAG}_bench_*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['value']), round(d['ms_per_step'],4), d['launch'], d.get('exchange',{}).get('form',''))
except Exception as e: print('$f', 'FAILED', e)"; done
cat gpurun_out/${TAG}_stage_step.txt gpurun_out/${TAG}_loops_step.txt
