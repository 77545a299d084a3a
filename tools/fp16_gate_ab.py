"""Price of the fp16 mode's erf GEGLU gate: run once per library build (E2V_LIB_PATH = the shipped .so, then the one built with
`make EXTRA=-DE2V_F16_FAST_GATE OUT=../lib/libeeg2video_hip_fastgate.so OBJDIR=../lib/obj_fg`, whose fp16 mode takes the bf16 mode's
logistic gate).  Prints (i) the distance of one full-size fp16 UNet forward from the fp32 HIP forward on the same input (the fp32 path
is within 1e-5 of the oracle), (ii) ms per guided DDIM step at B = 32 in fp16 and bf16 (interleaved, best of ROUNDS)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eeg2video_amd.pipeline import build_pipeline
from eeg2video_amd.weights import UNetConfig, VAEConfig, counter_normal
pipe = build_pipeline(UNetConfig(), VAEConfig(), device=0)
eng = pipe.unet.engine
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
x = t(counter_normal(1234, "latent", (1, 4, 6, 36, 64))).cuda()
cond = torch.cat([t(counter_normal(1236, "uncond", (1, 77, 768))), t(counter_normal(1235, "cond", (1, 77, 768)))]).cuda()
xx = torch.cat([x, x])
eps32 = eng.unet_forward(xx, [751], cond).clone()
rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
for mode in ("fp16", "bf16"):
    eng.set_compute_dtype(mode)
    e = eng.unet_forward(xx, [751], cond)
    print(f"{mode}: one UNet forward vs the fp32 HIP forward: max-abs / max-ref uncond {rel(e[:1], eps32[:1]):.3e} cond {rel(e[1:], eps32[1:]):.3e}")
B = int(os.environ.get("B", "32"))
lat = torch.stack([t(counter_normal(1234 + k, "latent", (4, 6, 36, 64))) for k in range(B)]).cuda()
cnd = torch.stack([t(counter_normal(1235 + 7919 * k, "cond", (77, 768))) for k in range(B)]).cuda()
unc = t(counter_normal(1236, "uncond", (1, 77, 768))).cuda()
best = {"fp16": 1e9, "bf16": 1e9}
for rnd in range(int(os.environ.get("ROUNDS", "3"))):
    for mode in ("fp16", "bf16"):
        eng.set_compute_dtype(mode)
        eng.generate(lat, cnd, unc, 1, 12.5, 0.0, decode=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.generate(lat, cnd, unc, 4, 12.5, 0.0, decode=False)
        torch.cuda.synchronize()
        best[mode] = min(best[mode], (time.perf_counter() - t0) / 4 * 1e3)
print(f"library {os.environ.get('E2V_LIB_PATH', '(shipped)')}: B = {B} guided DDIM step: fp16 {best['fp16']:.2f} ms, bf16 {best['bf16']:.2f} ms, fp16 / bf16 = {best['fp16'] / best['bf16']:.4f}")
