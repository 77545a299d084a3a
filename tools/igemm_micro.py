"""Micro-benchmark of the implicit-GEMM kernel on the UNet's dominant shapes (B = 8 clips -> 16 samples)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg2video_amd.engine import Engine
from eeg2video_amd.weights import TINY_UNET, TINY_VAE

eng = Engine(TINY_UNET, TINY_VAE, 0)
eng.set_compute_dtype(os.environ.get("DTYPE", "fp32"))
reps = int(os.environ.get("REPS", "5"))
only = os.environ.get("ONLY", "")
shapes = [
    # name, kind, n_img, H, W, Cin, Cout
    ("conv L0 320->320", "conv", 96, 36, 64, 320, 320),
    ("conv L0 640->320", "conv", 96, 36, 64, 640, 320),
    ("conv L1 640->640", "conv", 96, 18, 32, 640, 640),
    ("conv L2 1280->1280", "conv", 96, 9, 16, 1280, 1280),
    ("conv L3 1280->1280", "conv", 96, 5, 8, 1280, 1280),
    ("lin L0 320->320", "lin", 221184, 0, 0, 320, 320),
    ("lin L0 320->960", "lin", 221184, 0, 0, 320, 960),
    ("lin L1 640->1920", "lin", 55296, 0, 0, 640, 1920),
    ("lin L2 1280->1280", "lin", 13824, 0, 0, 1280, 1280),
    ("vae conv 288x512 128->128 (6 fr)", "conv", 6, 288, 512, 128, 128),
]
for name, kind, n, h, w, ci, co in shapes:
    if only and only not in name:
        continue
    if kind == "conv":
        x = torch.randn(n * h * w, ci, device="cuda")
        wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
        b = torch.randn(co, device="cuda")
        fn = lambda: eng.op_conv3x3(x, wt, b, n_img=n, Hs=h, Ws=w)
        flops = 2.0 * n * h * w * co * ci * 9
    else:
        x = torch.randn(n, ci, device="cuda")
        wt = torch.randn(co, ci, device="cuda") * 0.05
        b = torch.randn(co, device="cuda")
        fn = lambda: eng.op_linear(x, wt, b)
        flops = 2.0 * n * co * ci
    fn(); fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    # note: op_conv3x3 re-packs the weight per call (a small extra kernel); time only the main kernel by event pairs around fn
    ts = []
    for r in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    best = min(ts)
    print(f"{name:36s} {best*1e3:8.3f} ms  {flops/best/1e12:7.1f} TF (wall, incl. weight pack + launch)")
