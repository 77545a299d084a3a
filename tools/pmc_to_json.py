"""Fold two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) into profiles/pmc_dominant_kernel.json.

usage: python tools/pmc_to_json.py <fetch_dir> <write_dir> <out.json> "<command that was profiled>"

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are reported in KiB;
on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so reads are doubled; WRITE_SIZE is exact for 16-byte-per-lane
stores (which is what the igemm / Winograd / GroupNorm / LayerNorm kernels issue).  Values are averages per launch of a
kernel class over every launch in the run.
"""
import collections, csv, glob, json, os, sys

CLASSES = [("igemm_kernel<0, false>", "igemm_f32"), ("igemm_k16_kernel", "igemm_f32"), ("igemm_x3_kernel", "igemm_f32x3"), ("igemm_kernel<0, true>", "igemm_bf16"),
           ("flash_attn_bf16", "flash_attn_bf16"), ("flash_attn", "flash_attn"), ("wino_in", "wino_in"), ("wino4_in", "wino_in"), ("wino_out", "wino_out"), ("wino4_out", "wino_out"),
           ("gn_partial", "gn_partial"), ("gn_apply", "gn_apply"), ("layernorm", "layernorm"), ("temporal_attn", "temporal_attn")]


def classify(name):
    for pat, cls in CLASSES:
        if pat in name:
            return cls
    return None


def collect(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    per = collections.defaultdict(lambda: [set(), 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            cls = classify(r["Kernel_Name"])
            if cls:
                per[cls][0].add(r["Dispatch_Id"])
                per[cls][1] += float(r["Counter_Value"])
    return {k: (len(v[0]), v[1]) for k, v in per.items()}


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {}
for cls in fetch:
    n, fk = fetch[cls]
    nw, wk = write.get(cls, (n, 0.0))
    rd = fk * 1024.0 * 2.0 / n
    wr = wk * 1024.0 / max(nw, 1)
    out[cls] = {"launches": n, "fetch_size_kib_sum": fk, "write_size_kib_sum": wk,
                "hbm_read_bytes_per_launch_corrected_x2": rd, "hbm_write_bytes_per_launch": wr,
                "hbm_bytes_per_launch": rd + wr}
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `" + sys.argv[4] + "`; FETCH_SIZE doubled per "
                "MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE as is (exact for 16-B-per-lane stores). "
                "Average over all launches of the kernel class in the run.")
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "_note"}, indent=1))
