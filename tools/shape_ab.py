"""Same-process A/B of kernel variants per GEMM shape (DESIGN section 10 knobs): for each setting of the knobs, one warm-up DDIM step and
one event-instrumented step (+ decode) at batch B with shape-tagged kernel classes; prints the per-shape table side by side.

    DTYPE=bf16 python tools/shape_ab.py 32 E2V_BGEMM_T256=0 E2V_BGEMM_T256=1            # every argument after B is one variant: K=V[,K=V...]
"""
import json, os, re, sys
os.environ["E2V_PROFILE_DETAIL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eeg2video_amd.pipeline import build_pipeline
from eeg2video_amd.weights import UNetConfig, VAEConfig, counter_normal
pipe = build_pipeline(UNetConfig(), VAEConfig(), device=0)
eng = pipe.unet.engine
eng.set_compute_dtype(os.environ.get("DTYPE", "bf16"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
variants = sys.argv[2:] or ["E2V_BGEMM_T256=0", "E2V_BGEMM_T256=1"]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
lat = torch.stack([t(counter_normal(1234 + k, "latent", (4, 6, 36, 64))) for k in range(B)]).cuda()
cond = torch.stack([t(counter_normal(1235 + k, "cond", (77, 768))) for k in range(B)]).cuda()
unc = t(counter_normal(1236, "uncond", (1, 77, 768))).cuda()
DEC = os.environ.get("DECODE", "1") == "1"
ROUNDS = int(os.environ.get("ROUNDS", "2"))
norm = lambda k: re.sub(r" (T\d+|rb1=\d+|w\d+|s\d+|k16)(?= |$)", "", k)
tabs = [dict() for _ in variants]
for rnd in range(ROUNDS):                    # interleaved rounds: A B A B (cdna_hip_programming.md rule 24)
    for vi, var in enumerate(variants):
        for kv in var.split(","):
            k, v = kv.split("=")
            eng.set_knob(k, int(v))
        eng.generate(lat, cond, unc, 1, 12.5, 0.0, decode=DEC)
        torch.cuda.synchronize()
        eng.profile_begin()
        eng.generate(lat, cond, unc, 1, 12.5, 0.0, decode=DEC)
        tab = eng.profile_end()
        for k, v in tab.items():
            e = tabs[vi].setdefault(norm(k), {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0})
            for f in e:
                e[f] += v[f] / ROUNDS
keys = sorted(set().union(*[set(tb) for tb in tabs]), key=lambda k: -max(tb.get(k, {"ms": 0})["ms"] for tb in tabs))
print("variants:", variants)
for k in keys[:int(os.environ.get("TOP", "70"))]:
    ms = [tb.get(k, {"ms": float("nan")})["ms"] for tb in tabs]
    fl = next((tb[k]["flops"] for tb in tabs if k in tb), 0.0)
    n = next((tb[k]["launches"] for tb in tabs if k in tb), 0)
    tf = "  ".join(f"{m:8.3f} ms {fl / m / 1e9 if m == m and m > 0 else 0:7.1f} TF" for m in ms)
    ratio = f"{ms[-1] / ms[0]:5.2f}x" if len(ms) > 1 and ms[0] == ms[0] and ms[0] > 0 else ""
    print(f"{tf}  {ratio}  n={n:5.1f}  {k}")
tot = [sum(v["ms"] for v in tb.values()) for tb in tabs]
print("total ms per step (+ decode):", ["%.1f" % x for x in tot])
cls = lambda k: k.split(" ")[0]
for c in sorted(set(cls(k) for k in keys)):
    s = [sum(v["ms"] for k, v in tb.items() if cls(k) == c) for tb in tabs]
    f = [sum(v["flops"] for k, v in tb.items() if cls(k) == c) for tb in tabs]
    print(f"  class {c:34s}", "  ".join(f"{a:9.2f} ms {b / a / 1e9 if a > 0 else 0:7.1f} TF" for a, b in zip(s, f)))
json.dump({v: tb for v, tb in zip(variants, tabs)}, open(os.environ.get("OUT", "gpurun_out/shape_ab.json"), "w"), indent=1)
