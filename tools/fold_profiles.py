"""Copy the outputs of tools/profile_round.sh (gpurun_out/prof_<tag>/) into profiles/ under the round's names and rebuild
profiles/pmc_dominant_kernel.json (what bench.py quotes as roofline.traffic) from the two PMC summaries.

usage: python tools/fold_profiles.py r03          # expects gpurun_out/prof_r03_fp32_b8 and gpurun_out/prof_r03_bf16_b32 (prof_<tag>_fp16_b32 if present)
"""
import json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
dom = {}
for cfg, cls, batch, dtype in (("fp32_b8", "igemm_f32", 8, "fp32"), ("bf16_b32", "igemm_bf16", 32, "bf16"), ("fp16_b32", "igemm_fp16", 32, "fp16")):
    src = os.path.join(R, "gpurun_out", f"prof_{tag}_{cfg}")
    if cfg == "fp16_b32" and not os.path.isdir(src):
        continue
    for a, b in (("kernel_stats.csv", "rocprofv3_kernel_stats.csv"), ("kernel_classes_hip_events.json", "kernel_classes_hip_events.json"),
                 ("pmc_summary.json", "pmc_summary.json"), ("stats_bench.json", "bench_under_rocprof.json")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copy(os.path.join(src, a), os.path.join(R, "profiles", f"{tag}_{cfg}_{b}"))
        else:
            print("missing", os.path.join(src, a))
    s = json.load(open(os.path.join(src, "pmc_summary.json")))
    rec = dict(s[cls])
    rec.update(batch=batch, dtype=dtype, source_file=f"profiles/{tag}_{cfg}_pmc_summary.json")
    dom[cls] = rec
    dom.setdefault("_source", s["_source"].split(" over `")[0] + " over `bench.py --steps 1 --warmup 0` of the named batch / dtype (tools/profile_round.sh, SYNC=1); "
                   "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B), WRITE_SIZE as is; averages over all launches of the class")
json.dump(dom, open(os.path.join(R, "profiles", "pmc_dominant_kernel.json"), "w"), indent=1, sort_keys=True)
for cls, r in dom.items():
    if not cls.startswith("_"):
        print(cls, "hbm bytes/launch %.3e" % r["hbm_bytes_per_launch"], "mfma pipe %.3f" % r.get("mfma_pipe_utilisation", float("nan")), "launches", r["launches"])
