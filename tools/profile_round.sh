#!/bin/bash
# One gpurun call: the rocprofv3 evidence of a round for one bench configuration -- kernel-trace stats of the timed command
# and separate PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy) over the same workload, written under gpurun_out/ and folded
# into profiles/ by the caller.   usage: tools/profile_round.sh <tag> [bench.py args ...]      e.g. r02_fp32_b8 / r02_bf16_b32 --dtype bf16
#
# The kernel-trace pass runs the workload AS IT IS TIMED (every launch queued asynchronously).  The counter passes are run with
# SYNC=1 (E2V_SYNC_EACH_STEP=1: the stream is drained after every DDIM step, <= ~680 launches in flight instead of a whole pass of
# ~34 000): with the deep queue rocprofiler-sdk's queue-intercept callback faulted (profiles/r02_pmc_async_abort_README.md names
# the frames).  Per-dispatch counters do not depend on queue depth.  Every pass keeps its stderr, a failing pass is recorded
# and NOT retried, and the remaining passes of the call are skipped.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
export PYTHONFAULTHANDLER=1
OUT=$R/gpurun_out/prof_$TAG
if [ "${SKIP_STATS:-0}" != "1" ]; then rm -rf $OUT; fi
mkdir -p $OUT
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-configs2 $*"
if [ "${SKIP_STATS:-0}" != "1" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/bench.py $ARGS --kernel-table $OUT/kernel_classes_hip_events.json > $OUT/stats_bench.json 2> $OUT/stats.err
echo "stats pass rc=$?"
fi
PARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-configs2 $*"
export E2V_LOG_MAPS=1
# SYNC=1: drain the stream after every DDIM step in the counter passes (see the header)
if [ "${SYNC:-0}" = "1" ]; then export E2V_SYNC_EACH_STEP=1; fi
if [ "${SKIP_STATS:-0}" = "1" ]; then echo "stats pass skipped"; fi
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    name=${pass%%:*}; ctrs=${pass#*:}
    rocprofv3 --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o pmc -- python3 $R/bench.py $PARGS > $OUT/pmc_${name}_bench.json 2> $OUT/pmc_$name.err
    rc=$?
    echo "pmc $name pass rc=$rc"
    if [ $rc -ne 0 ]; then
        echo "pmc $name pass FAILED (rc=$rc): stderr kept in $OUT/pmc_$name.err; no retry, skipping the remaining passes"
        tail -n 40 $OUT/pmc_$name.err
        break
    fi
done
cd $R
python3 tools/pmc_to_json.py $OUT $OUT/pmc_summary.json "bench.py $PARGS" "${E2V_SYNC_EACH_STEP:-0}" || true
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*.csv" ! -name "kernel_stats.csv" -size +2M -delete
ls -la $OUT
