"""Print the parts of a bench.py JSON line that matter when checking a run by eye.  usage: python tools/show_bench.py <file.json>"""
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup", "dtype")})
if d.get("roofline"):
    print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "achieved", "frac", "avg_launch_us", "share_of_gpu_time", "sample", "traffic")})
c = d.get("configs2")
if c and "value" in c:
    print("configs2", {k: c[k] for k in ("value", "ms_per_step", "steps", "warmup")}, {k: c["roofline"].get(k) for k in ("kernel", "achieved", "frac", "sample")})
elif c:
    print("configs2", c)
if d.get("cpu_baseline"):
    print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "wall_s", "cores", "overlapped_with", "skipped")})
print("parity", d.get("parity"), "gpu_over_cpu", d.get("gpu_over_cpu"))
