#!/bin/bash
# rocprofv3 counter passes over the implicit-GEMM micro-benchmark (one shape): where do the waves of the dominant kernel spend
# their cycles?  usage: tools/pmc_micro.sh <tag> <DTYPE> <SCALE> "<ONLY substring>"
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; export DTYPE=$2; export SCALE=$3; export ONLY="$4"; export REPS=2
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
run() {   # name, counters
    rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/$1 -o p -- python3 $R/${MICRO:-tools/igemm_micro.py} > $OUT/$1.log 2>&1 || echo "pass $1 failed"
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU"
run b "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES"
run c "GRBM_GUI_ACTIVE FETCH_SIZE"
run d "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
cd $R
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, set()]))
dur = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
for f in glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        d = dur[k]; d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); d[1] += 1
res = {}
for k, cs in agg.items():
    if not any(s in k for s in ("gemm", "attn", "norm", "gn_")):
        continue
    res[k] = {c: v[0] / max(len(v[1]), 1) for c, v in cs.items()}
    if dur[k][1]:
        res[k]["avg_ns_profiled"] = dur[k][0] / dur[k][1]
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
PY
find $OUT -name "*.csv" -size +4M -delete
