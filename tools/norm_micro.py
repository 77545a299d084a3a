"""Micro-benchmark of the bf16 GroupNorm(+SiLU) and LayerNorm passes at the UNet's B = 32 shapes (64 CFG samples): the flat-index apply
pass against the row-tiled one (E2V_GN_ROWS) and rows per workgroup of the two passes (E2V_GN_CHUNK_ROWS), switched inside one process.  (Runs of samples sized for the
Infinity Cache, E2V_GN_GROUP_MB, measured slower: profiles/r03_norm_micro.log.)  usage: python tools/norm_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg2video_amd.engine import Engine
from eeg2video_amd.weights import TINY_UNET, TINY_VAE
eng = Engine(TINY_UNET, TINY_VAE, 0)
eng.set_compute_dtype("bf16")
n = int(os.environ.get("N", "64"))
variants = [{"E2V_GN_ROWS": 1, "E2V_GN_CHUNK_ROWS": 256, "E2V_GN_CHUNK_ROWS_SMALL": 256}, {"E2V_GN_ROWS": 1, "E2V_GN_CHUNK_ROWS": 256, "E2V_GN_CHUNK_ROWS_SMALL": 128},
            {"E2V_GN_ROWS": 1, "E2V_GN_CHUNK_ROWS": 256, "E2V_GN_CHUNK_ROWS_SMALL": 64}, {"E2V_GN_ROWS": 1, "E2V_GN_CHUNK_ROWS": 128, "E2V_GN_CHUNK_ROWS_SMALL": 64},
            {"E2V_GN_ROWS": 0, "E2V_GN_CHUNK_ROWS": 256, "E2V_GN_CHUNK_ROWS_SMALL": 64}]
shapes = [("L0 320", 6 * 2304, 320, 0), ("L0 320+320", 6 * 2304, 320, 320), ("L0 640+320", 6 * 2304, 640, 320), ("L1 640", 6 * 576, 640, 0),
          ("L1 1280+640", 6 * 576, 1280, 640), ("L2 1280", 6 * 144, 1280, 0), ("L2 1280+1280", 6 * 144, 1280, 1280)]
for name, P, c0, c1 in shapes:
    x0 = torch.randn(n * P, c0, device="cuda")
    x1 = torch.randn(n * P, c1, device="cuda") if c1 else None
    g = torch.rand(c0 + c1, device="cuda") + 0.5
    b = torch.randn(c0 + c1, device="cuda") * 0.1
    ref = None
    line = f"{name:14s} n={n}:"
    for v in variants:
        for k, val in v.items():
            eng.set_knob(k, val)
        best = 1e9
        for _ in range(4):
            eng.profile_begin()
            y = eng.op_groupnorm(x0, g, b, samples=n, P=P, groups=32, eps=1e-5, silu=True, x1=x1)
            pr = eng.profile_end()
            best = min(best, pr["groupnorm_silu"]["ms"])
        if ref is None:
            ref = y
        err = (y - ref).abs().max().item()
        byt = 2.0 * 2 * n * P * (c0 + c1)
        line += f"  [{v['E2V_GN_ROWS']},{v['E2V_GN_CHUNK_ROWS']:3d},{v['E2V_GN_CHUNK_ROWS_SMALL']:3d}] {best:.3f} ms {byt / best / 1e6:5.0f} GB/s (d {err:.1e})"
        del y
    print(line, flush=True)
    del x0, x1, ref
for k, v in (("E2V_GN_ROWS", 1), ("E2V_GN_CHUNK_ROWS", 256), ("E2V_GN_CHUNK_ROWS_SMALL", 64)):
    try:
        eng.set_knob(k, v)
    except ValueError:      # E2V_GN_ROWS exists in `make AB=1` builds only
        pass
for name, rows, c in [("LN L0 320", n * 6 * 2304, 320), ("LN L1 640", n * 6 * 576, 640), ("LN L2 1280", n * 6 * 144, 1280)]:
    x = torch.randn(rows, c, device="cuda")
    g = torch.rand(c, device="cuda") + 0.5
    b = torch.randn(c, device="cuda") * 0.1
    best = 1e9
    for _ in range(4):
        eng.profile_begin()
        y = eng.op_layernorm(x, g, b)
        pr = eng.profile_end()
        best = min(best, pr["layernorm"]["ms"])
    print(f"{name:14s}: {best:.3f} ms {2.0 * 2 * rows * c / best / 1e6:5.0f} GB/s", flush=True)
    del x, y
eng.set_compute_dtype("fp32")
