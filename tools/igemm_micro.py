"""Micro-benchmark of the implicit-GEMM kernel on the UNet's dominant shapes (B = 8 clips -> 16 samples)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg2video_amd.engine import Engine
from eeg2video_amd.weights import TINY_UNET, TINY_VAE

eng = Engine(TINY_UNET, TINY_VAE, 0)
if "DTYPE" in os.environ:
    eng.set_compute_dtype(os.environ["DTYPE"])
eng.set_conv_algo(os.environ.get("ALGO", "direct"))
reps = int(os.environ.get("REPS", "5"))
only = os.environ.get("ONLY", "")
scale = int(os.environ.get("SCALE", "1"))      # 4 = the B = 32 shapes of BASELINE configs[2]
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
    n *= scale
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
    # op_conv3x3 re-packs the weight per call (a small extra kernel): report the igemm launch alone, from the library's
    # own HIP-event profiler (same events bench.py's roofline object uses)
    best = 1e9
    for r in range(reps):
        eng.profile_begin()
        fn()
        prof = eng.profile_end()
        tot = sum(k["ms"] for nm, k in prof.items() if nm.startswith(("igemm", "wino_"))) * 1e-3
        parts = {nm: round(k["ms"], 3) for nm, k in prof.items() if nm.startswith(("igemm", "wino_"))}
        best = min(best, tot)
    print(f"{name:36s} {best*1e3:8.3f} ms  {flops/best/1e12:7.1f} TF (conv kernels alone, HIP events) {parts if len(parts) > 1 else ''}")
