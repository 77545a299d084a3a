"""Micro-benchmark of the sparse-causal attention kernel at the UNet's level-0..2 shapes (B = 8 -> 16 samples)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg2video_amd.engine import Engine
from eeg2video_amd.weights import TINY_UNET, TINY_VAE
eng = Engine(TINY_UNET, TINY_VAE, 0)
reps = int(os.environ.get("REPS", "3"))
for name, n, f, nq, d in [("L0 d40", 16, 6, 2304, 40), ("L1 d80", 16, 6, 576, 80), ("L2 d160", 16, 6, 144, 160)]:
    heads = 8; c = heads * d
    qkv = torch.randn(n * f * nq, 3 * c, device="cuda")
    fn = lambda: eng.op_attention(qkv[:, :c], qkv[:, c:2*c], qkv[:, 2*c:], n=n, F=f, heads=heads, D=d, Nq=nq, Nk=nq, mode=0, scale=d ** -0.5)
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    alg = 4.0 * n * f * heads * nq * (2 * nq) * d
    print(f"{name}: {min(ts)*1e3:.3f} ms  {alg/min(ts)/1e12:.1f} TF algorithmic")
