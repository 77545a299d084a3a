"""f32x3 GEMM micro-benchmark on long-K / big-M shapes (steady-state main loop)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg2video_amd.engine import Engine
from eeg2video_amd.weights import TINY_UNET, TINY_VAE
eng = Engine(TINY_UNET, TINY_VAE, 0)
for name, m, k, n in [("M55296 K2560 N640", 55296, 2560, 640), ("M221184 K1280 N320", 221184, 1280, 320), ("M65536 K1280 N1280", 65536, 1280, 1280),
                      ("M221184 K320 N960", 221184, 320, 960)]:
    x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.05; b = torch.randn(n, device="cuda")
    best = 1e9
    for _ in range(4):
        eng.profile_begin(); eng.op_linear(x, w, b); pr = eng.profile_end()
        best = min(best, sum(v["ms"] for kk, v in pr.items() if kk.startswith("igemm")))
    print(f"{name}: {best:.3f} ms  {2.0*m*n*k/best/1e9:.1f} TF")
