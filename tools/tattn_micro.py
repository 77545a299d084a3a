"""Micro-benchmark of the temporal attention kernel at the UNet's level-0..2 shapes, fp32 rows (B = 8 -> 16 samples) and bf16 rows
(B = 32 -> 64 samples), the wave-per-pixel kernel against the LDS-staged one (E2V_TATTN_WAVE=0 in a second process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg2video_amd.engine import Engine
from eeg2video_amd.weights import TINY_UNET, TINY_VAE
eng = Engine(TINY_UNET, TINY_VAE, 0)
print("E2V_TATTN_WAVE =", os.environ.get("E2V_TATTN_WAVE", "1"))
for mode, n in (("fp32", 16), ("bf16", 64)):
    eng.set_compute_dtype(mode)
    for name, f, hw, d in [("L0 d40", 6, 2304, 40), ("L1 d80", 6, 576, 80), ("L2 d160", 6, 144, 160)]:
        heads = 8; c = heads * d
        qkv = torch.randn(n * f * hw, 3 * c, device="cuda")
        best = 1e9
        for _ in range(5):
            eng.profile_begin()
            eng.op_temporal_attention(qkv, n=n, F=f, HW=hw, heads=heads, D=d, scale=d ** -0.5)
            pr = eng.profile_end()
            best = min(best, pr["temporal_attn"]["ms"])
        byt = (4.0 if mode == "fp32" else 2.0) * 4 * n * f * hw * c
        print(f"{mode} {name} n={n}: {best:.3f} ms  {byt/best/1e6:.0f} GB/s algorithmic")
eng.set_compute_dtype("fp32")
