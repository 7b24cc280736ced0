#!/bin/bash
# one gpurun call: kernel-trace stats + the two PMC passes + the default bench line  ->  gpurun_out/r03_*   (tag = $1, default r03)
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
ARGS="bench.py --no-cpu-baseline --launch eager"
LS2FM_SERIAL=1 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_kt -- python $ARGS --steps 100 --warmup 10 > gpurun_out/${TAG}_kt_bench.json 2>/dev/null
python tools/prof_top.py gpurun_out/${TAG}_kt 24 > gpurun_out/${TAG}_rocprof_kernel_stats.txt
LS2FM_SERIAL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch -- python $ARGS --steps 20 --warmup 5 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/${TAG}_fetch > gpurun_out/${TAG}_pmc_fetch.txt
LS2FM_SERIAL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${TAG}_write -- python $ARGS --steps 20 --warmup 5 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/${TAG}_write > gpurun_out/${TAG}_pmc_write.txt
rm -rf gpurun_out/${TAG}_kt gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
tail -c 600 gpurun_out/${TAG}_bench_default.json
