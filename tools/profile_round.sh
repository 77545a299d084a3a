#!/bin/bash
# One gpurun call: rocprofv3 kernel stats + the two PMC passes of the bench's event-instrumented workload mix
# (2 DDIM steps + decode per pass, B = 8), written under gpurun_out/ and folded into profiles/ by the caller.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
ARGS="--ddim-steps 2 --profile-ddim-steps 2 --steps 1 --warmup 1 --no-cpu-baseline"
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o stats -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_stats_bench.json 2> $R/gpurun_out/prof_stats.err
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o pmc -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_fetch_bench.json 2> $R/gpurun_out/pmc_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o pmc -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_write_bench.json 2> $R/gpurun_out/pmc_write.err
echo "write pass done"
cd $R
python3 tools/pmc_to_json.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_dominant_kernel.json "bench.py $ARGS"
find gpurun_out/prof_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
# the raw traces are large: keep only the summaries
find gpurun_out/prof_stats gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" ! -name "*kernel_stats.csv" -size +8M -delete
