#!/bin/bash
# One gpurun call: rocprofv3 kernel stats + the two PMC passes of the bench's event-instrumented workload mix
# (50 DDIM steps + decode per pass, B = 8), written under gpurun_out/ and folded into profiles/ by the caller.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline"          # the default workload: 50-step passes, the instrumented one included
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_stats_bench.json 2> $R/gpurun_out/prof_stats.err
echo "stats pass done"
# rocprofv3 7.2 --pmc segfaults inside its dispatch hook once ~8k launches are queued behind each other (a 50-step pass
# queues 45k): E2V_SYNC_EACH_STEP=1 drains the stream after every DDIM step in the counter passes (same kernels, same mix)
PARGS="--steps 1 --warmup 0 --no-cpu-baseline"
export E2V_SYNC_EACH_STEP=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o pmc -- python3 $R/bench.py $PARGS > $R/gpurun_out/pmc_fetch_bench.json 2> $R/gpurun_out/pmc_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o pmc -- python3 $R/bench.py $PARGS > $R/gpurun_out/pmc_write_bench.json 2> $R/gpurun_out/pmc_write.err
echo "write pass done"
unset E2V_SYNC_EACH_STEP
cd $R
python3 tools/pmc_to_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_dominant_kernel.json "bench.py $PARGS"
find gpurun_out/prof_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
# the raw traces are large: keep only the summaries
find gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" ! -name "*kernel_stats.csv" -size +8M -delete
