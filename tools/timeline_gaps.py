"""Where does the wall time of a launch-bound pass go?  Fold a `rocprofv3 --kernel-trace` CSV into: GPU-busy time vs. the span of
the trace, the idle time between consecutive kernels (histogram), and per kernel class the launches, the busy time and the idle
time that FOLLOWS a launch of the class (the gap a dependent successor waits through: queue barrier + dispatch + ramp).

usage: python tools/timeline_gaps.py <dir with *kernel_trace.csv> [out.json]

Used for the B = 1 question of DESIGN 3.9 (about 400 dependent launches per 14 ms DDIM step): is the step bounded by the kernels or by
what lies between them?
"""
import collections, csv, glob, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_to_json import classify  # noqa: E402  (the same kernel classes as the PMC summaries)


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    if not rows:
        raise SystemExit("no kernel_trace.csv under " + sys.argv[1])
    # the steady part: skip everything before the LAST long idle stretch > 50 ms (weight upload, warm-up pass boundaries keep their own)
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    gaps = []
    per = collections.defaultdict(lambda: [0, 0, 0])        # class -> launches, busy ns, idle ns after
    pairs = collections.defaultdict(lambda: [0, 0])         # (class before, class after) of a gap > 2 us -> count, ns
    short = lambda n: classify(n) or n.split("(")[0].split("<")[0][-40:]
    frontier = rows[0][1]
    for i, (s, e, name) in enumerate(rows):
        cls = classify(name) or name.split("(")[0].split("<")[0][-40:]
        p = per[cls]
        p[0] += 1
        p[1] += e - s
        if i + 1 < len(rows):
            g = rows[i + 1][0] - max(frontier, e)
            frontier = max(frontier, e)
            if g < 20_000_000:                                 # (longer: host-side phases between passes, not launch gaps)
                gaps.append(max(g, 0))
                p[2] += max(g, 0)
            if g > 2000:
                q = pairs[(cls, short(rows[i + 1][2]), "long" if g >= 20_000_000 else "gap")]
                q[0] += 1
                q[1] += g
    gaps_sorted = sorted(gaps)
    hist = collections.Counter()
    for g in gaps:
        hist["<1us" if g < 1000 else "1-2us" if g < 2000 else "2-4us" if g < 4000 else "4-8us" if g < 8000 else "8-16us" if g < 16000
             else "16-100us" if g < 100000 else ">100us"] += 1
    out = {
        "kernels": len(rows), "span_ms": span / 1e6, "busy_ms": busy / 1e6, "launch_gap_ms": sum(gaps) / 1e6,
        "gap_median_us": gaps_sorted[len(gaps) // 2] / 1e3, "gap_mean_us": sum(gaps) / len(gaps) / 1e3,
        "gap_p90_us": gaps_sorted[int(len(gaps) * 0.9)] / 1e3, "gap_histogram": dict(hist),
        "classes": {k: {"launches": v[0], "busy_ms": v[1] / 1e6, "avg_us": v[1] / v[0] / 1e3, "idle_after_ms": v[2] / 1e6,
                        "idle_after_avg_us": v[2] / v[0] / 1e3} for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:24]},
    }
    # the steady window: the last 100 DDIM steps of the trace (two passes of a 50-step run), from the end of the step before them
    ddim = [i for i, r in enumerate(rows) if "ddim_cfg_step" in r[2]]
    if len(ddim) > 100:
        i0, i1 = ddim[-101], ddim[-1]
        w = rows[i0 + 1:i1 + 1]
        t0, t1 = rows[i0][1], rows[i1][1]
        wbusy, wgap, front = 0, 0, t0
        big = []
        detail, prev = [], rows[i0][2]
        for s_, e_, n_ in w:
            if s_ > front:
                wgap += s_ - front
                if s_ - front > 20000:
                    big.append((s_ - front) / 1e3)
                    detail.append({"idle_us": (s_ - front) / 1e3, "at_ms": (front - t0) / 1e6, "after": short(prev), "before": short(n_)})
            wbusy += e_ - s_
            front = max(front, e_)
            prev = n_
        out["last_100_ddim_steps"] = {"span_ms": (t1 - t0) / 1e6, "ms_per_step": (t1 - t0) / 1e8, "busy_ms": wbusy / 1e6, "idle_ms": wgap / 1e6,
                                      "kernels_per_step": len(w) / 100.0, "idle_stretches_over_20us": len(big), "their_sum_ms": sum(big) / 1e3,
                                      "note": "includes the decode and the host work between the two passes",
                                      "stretches": sorted(detail, key=lambda d: -d["idle_us"])[:60]}
    out["gaps_over_2us_by_neighbours"] = [{"after": k[0], "before": k[1], "kind": k[2], "count": v[0], "total_ms": v[1] / 1e6, "avg_us": v[1] / v[0] / 1e3}
                                           for k, v in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:30]]
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
