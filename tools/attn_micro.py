"""Micro-benchmark of the sparse-causal attention kernel at the UNet's level-0..2 shapes: fp32 (B = 8 -> 16 samples) and bf16 rows
(B = 32 -> 64 samples), HIP-event kernel time; the bf16 variants (E2V_ATTN_Q64: 64 queries per wave, 2 = also at d = 80; E2V_ATTN_FOLD,
E2V_ATTN_KT64 with OLD_VARIANTS=1) A/B'd in one process, interleaved."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg2video_amd.engine import Engine
from eeg2video_amd.weights import TINY_UNET, TINY_VAE
eng = Engine(TINY_UNET, TINY_VAE, 0)
reps = int(os.environ.get("REPS", "4"))
shapes = [("L0 d40", 6, 2304, 40), ("L1 d80", 6, 576, 80), ("L2 d160", 6, 144, 160)]
def run(mode, n, variants):
    eng.set_compute_dtype(mode)
    for name, f, nq, d in shapes:
        heads = 8; c = heads * d
        qkv = torch.randn(n * f * nq, 3 * c, device="cuda")
        best = {v: 1e9 for v in variants}
        for _ in range(reps):
            for v in variants:
                for kv in v.split(","):
                    if kv:
                        k, val = kv.split("="); eng.set_knob(k, int(val))
                eng.profile_begin()
                eng.op_attention(qkv[:, :c], qkv[:, c:2*c], qkv[:, 2*c:], n=n, F=f, heads=heads, D=d, Nq=nq, Nk=nq, mode=0, scale=d ** -0.5)
                pr = eng.profile_end()
                ms = [x["ms"] for kx, x in pr.items() if kx.startswith("flash_attn")][0]
                best[v] = min(best[v], ms)
        alg = 4.0 * n * f * heads * nq * (2 * nq) * d
        print(f"{mode} {name} n={n}: " + "  ".join(f"[{v or 'default'}] {best[v]:.3f} ms {alg/best[v]/1e9:.0f} TF" for v in variants))
run("fp32", 16, [""])
if os.environ.get("OLD_VARIANTS"):
    run("bf16", 64, ["E2V_ATTN_Q64=0,E2V_ATTN_FOLD=0,E2V_ATTN_KT64=0", "E2V_ATTN_Q64=0,E2V_ATTN_FOLD=1,E2V_ATTN_KT64=0", "E2V_ATTN_Q64=0,E2V_ATTN_FOLD=0,E2V_ATTN_KT64=1"])   # (AB=1 build)
try:                                   # `make AB=1` builds carry the other arms (phase form, plain softmax, forced workgroup sizes)
    eng.set_knob("E2V_ATTN_Q64P", 1); AB = True
except ValueError:
    AB = False
run("bf16", 64, ["E2V_ATTN_Q64=0", "E2V_ATTN_Q64=1"] + (["E2V_ATTN_Q64=1,E2V_ATTN_Q64P=0"] if AB else []) + [v for v in os.environ.get("EXTRA_VARIANTS", "").split(";") if v])
eng.set_knob("E2V_ATTN_Q64", 1)
if AB:
    eng.set_knob("E2V_ATTN_Q64P", 1); eng.set_knob("E2V_ATTN_Q64_NW", 0)
eng.set_compute_dtype("fp32")
