"""Semantic Predictor at the reference's size (310 -> 10000 x4 -> 77*768, 0.89 G parameters) at batch 1 .. 3: time per call and
weight bytes per second of the weight-streaming GEMV path (E2V_SEM_GEMV=0: the MFMA tile path), fp32 and bf16, with the result
checked against a plain torch evaluation of the same MLP on the GPU (fp64 accumulate)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eeg2video_amd.engine import Engine
from eeg2video_amd.semantic import CLIP
from eeg2video_amd.weights import SemanticConfig, TINY_UNET, TINY_VAE, UNetConfig

scfg = SemanticConfig()
eng = Engine(TINY_UNET.__class__(block_out_channels=TINY_UNET.block_out_channels, sample_size=8, cross_attention_dim=768), TINY_VAE, 0, sem_cfg=scfg)
g = torch.Generator().manual_seed(0)
dims = [310, 10000, 10000, 10000, 10000, 77 * 768]
sd = {}
for i in range(5):
    sd[f"mlp.{2 * i}.weight"] = (torch.randn(dims[i + 1], dims[i], generator=g) * (1.0 / dims[i] ** 0.5)).numpy()
    sd[f"mlp.{2 * i}.bias"] = (torch.randn(dims[i + 1], generator=g) * 0.1).numpy()
model = CLIP(scfg, engine=eng).load_state_dict({"state_dict": sd})
wbytes = sum(v.size for k, v in sd.items() if k.endswith("weight")) * 4
for dtype in ("fp32", "bf16"):
    eng.set_compute_dtype(dtype)
    for B in (1, 2, 3):
        x = torch.randn(B, 310, generator=g)
        y = model(x.cuda())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            y = model(x.cuda())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        ref = x.cuda().double()
        rb = (lambda t: t.to(torch.bfloat16).double()) if dtype == "bf16" else (lambda t: t.double())
        for i in range(5):
            w, b = torch.from_numpy(sd[f"mlp.{2 * i}.weight"]).cuda(), torch.from_numpy(sd[f"mlp.{2 * i}.bias"]).cuda()
            ref = rb(ref.float()) @ rb(w).T + b.double()
            if i < 4:
                ref = ref.relu()
        err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
        wb = wbytes / (2 if dtype == "bf16" else 1)
        print(f"{dtype} B={B}: {dt * 1e3:7.3f} ms per call, {wb / dt / 1e9:7.0f} GB/s of weights, max-abs/max-ref vs torch fp64 {err:.2e}")
eng.set_compute_dtype("fp32")
