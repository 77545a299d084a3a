"""Micro-benchmark of the HBM-bound kernels of the bf16-activation mode at the level-0 shapes of B = 32 (or SCALE / 4 of it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg2video_amd.engine import Engine
from eeg2video_amd.weights import TINY_UNET, TINY_VAE
eng = Engine(TINY_UNET, TINY_VAE, 0)
eng.set_compute_dtype(os.environ.get("DTYPE", "bf16"))
n = int(os.environ.get("SAMPLES", "16"))
F, HW, C = 6, 2304, 320
rows = n * F * HW
x = torch.randn(rows, C, device="cuda")
g, b = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
qkv = torch.randn(rows, 3 * C, device="cuda")
for _ in range(2):
    eng.op_groupnorm(x, g, b, samples=n, P=F * HW, groups=32, eps=1e-5, silu=True)
    eng.op_groupnorm(x, g, b, samples=n * F, P=HW, groups=32, eps=1e-6, silu=False)
    eng.op_layernorm(x, g, b)
    eng.op_temporal_attention(qkv, n=n, F=F, HW=HW, heads=8, D=40, scale=40 ** -0.5)
torch.cuda.synchronize()
eng.profile_begin()
eng.op_groupnorm(x, g, b, samples=n, P=F * HW, groups=32, eps=1e-5, silu=True)
eng.op_layernorm(x, g, b)
eng.op_temporal_attention(qkv, n=n, F=F, HW=HW, heads=8, D=40, scale=40 ** -0.5)
for k, v in eng.profile_end().items():
    print(f"{k:20s} {v['ms']:8.3f} ms  {v['bytes'] / v['ms'] / 1e6:8.0f} GB/s algorithmic")
