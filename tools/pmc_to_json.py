"""Fold the `rocprofv3 --pmc` passes of tools/profile_round.sh into one JSON: per kernel class, average per launch.

usage: python tools/pmc_to_json.py <prof_dir> <out.json> "<command that was profiled>"

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB;
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so reads are doubled; WRITE_SIZE is exact for 16-byte-per-lane
stores (which is what the GEMM / Winograd / GroupNorm / LayerNorm kernels issue).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles
summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
"""
import collections, csv, glob, json, os, sys

CLASSES = [("bgemm", "igemm_bf16"), ("igemm_kernel", "igemm_f32"), ("igemm_k16_kernel", "igemm_f32"), ("igemm_x3_kernel", "igemm_f32x3"),
           ("flash_attn_b16io", "flash_attn_bf16"), ("cross_attn_resident", "cross_attn_bf16"), ("flash_attn", "flash_attn"), ("wino4_in", "wino_in"), ("wino_in", "wino_in"),
           ("wino4_out", "wino_out"), ("wino_out", "wino_out"), ("gn_partial", "gn_partial"), ("gn_apply", "gn_apply"),
           ("layernorm", "layernorm"), ("temporal_attn", "temporal_attn")]


def classify(name):
    for pat, cls in CLASSES:
        if pat in name:
            # (round 5: the 16-bit kernels are templates over the element type -- the fp16 instances carry `_Float16` in their demangled and
            # `IDF16_` in their mangled names; bf16: `__bf16` / `IDF16b`)
            return cls.replace("bf16", "fp16") if cls.endswith("bf16") and ("_Float16" in name or "IDF16_" in name) else cls
    return None


def collect(d):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [set(), 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            cls = classify(r["Kernel_Name"])
            if cls:
                e = per[cls][r["Counter_Name"]]
                e[0].add(r["Dispatch_Id"])
                e[1] += float(r["Counter_Value"])
    return per


def main():
    per = collect(sys.argv[1])
    out = {}
    for cls, ctrs in per.items():
        o = {}
        for c, (ids, total) in ctrs.items():
            o[c + "_per_launch"] = total / max(len(ids), 1)
            o["launches"] = max(o.get("launches", 0), len(ids))
        if "FETCH_SIZE_per_launch" in o:
            o["hbm_read_bytes_per_launch_corrected_x2"] = o["FETCH_SIZE_per_launch"] * 1024.0 * 2.0
        if "WRITE_SIZE_per_launch" in o:
            o["hbm_write_bytes_per_launch"] = o["WRITE_SIZE_per_launch"] * 1024.0
        if "hbm_read_bytes_per_launch_corrected_x2" in o and "hbm_write_bytes_per_launch" in o:
            o["hbm_bytes_per_launch"] = o["hbm_read_bytes_per_launch_corrected_x2"] + o["hbm_write_bytes_per_launch"]
        if "SQ_VALU_MFMA_BUSY_CYCLES_per_launch" in o and o.get("GRBM_GUI_ACTIVE_per_launch"):
            # MFMA pipe utilisation: busy cycles per SIMD (1024 SIMDs) over the kernel's active cycles (GRBM_GUI_ACTIVE / 8 XCDs)
            o["mfma_pipe_utilisation"] = (o["SQ_VALU_MFMA_BUSY_CYCLES_per_launch"] / 1024.0) / (o["GRBM_GUI_ACTIVE_per_launch"] / 8.0)
        out[cls] = o
    out["_source"] = ("rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; one pass each) over `" + sys.argv[3] +
                      "`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE as is (exact for 16-B-per-lane stores); "
                      "averages over all launches of the kernel class in the run; E2V_SYNC_EACH_STEP=" + (sys.argv[4] if len(sys.argv) > 4 else "0") +
                      (" (the stream drained after every DDIM step: <= ~680 launches in flight)" if len(sys.argv) > 4 and sys.argv[4] == "1"
                       else " (every launch queued asynchronously as in the timed run)"))
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
