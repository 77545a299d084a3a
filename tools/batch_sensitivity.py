"""Fold the bench lines of a batch sweep (`bench.py --dtype D --batch B ...` written as <dir>/<dtype>_b<B>.json) into
profiles/r05_batch_sensitivity.json: clips/s, seconds per clip and the dominant class's rate per (dtype, batch), next to the round-4
figures the review quoted (B = 1 1.07 / 0.38 clips/s bf16 / fp32)."""
import glob, json, os, re, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r5y"
out = {"_source": f"bench.py --dtype D --batch B --steps 3 --warmup 1 --no-cpu-baseline on one MI355X box ({src}); 50-step DDIM, CFG 12.5, 6x288x512 decode, "
                  "frames copied to the host inside the timed step",
       "round4": {"bf16": {"1": 1.07, "2": 1.72, "4": 2.35, "32": 3.25}, "fp32": {"1": 0.38, "8": 0.585}}, "round5": {}}
for f in sorted(glob.glob(os.path.join(src, "*_b*.json"))):
    m = re.match(r"(\w+?)_b(\d+)\.json", os.path.basename(f))
    if not m:
        continue
    d = json.load(open(f))
    r = d.get("roofline") or {}
    out["round5"].setdefault(m.group(1), {})[m.group(2)] = {
        "clips_per_s": round(d["value"], 4), "s_per_clip": round(d["ms_per_step"] / 1e3 / int(m.group(2)), 4),
        "dominant": r.get("kernel"), "dominant_tflops": round(r.get("achieved", 0.0), 1), "dominant_share": round(r.get("share_of_gpu_time", 0.0), 3)}
json.dump(out, open("profiles/r05_batch_sensitivity.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out["round5"], indent=1))
