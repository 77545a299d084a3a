"""Full-size end-to-end deviation of the conv algorithms: one clip, 50-step DDIM + decode, same inputs, on the GPU.
direct implicit GEMM (the reference arithmetic) vs auto (F(2x2,3x3) where wide) vs auto + F(4x4,3x3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eeg2video_amd.weights import UNetConfig, VAEConfig, counter_normal

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lat = torch.stack([t(counter_normal(1234 + k, "latent", (4, 6, 36, 64))) for k in range(B)]).cuda()
cond = torch.stack([t(counter_normal(1235 + k, "cond", (77, 768))) for k in range(B)]).cuda()
unc = t(counter_normal(1236, "uncond", (1, 77, 768))).cuda()
res = {}
VARIANTS = [("direct", {"E2V_CONV_ALGO": "1"}), ("auto (F4 > F2 > direct)", {}), ("auto without F4", {"E2V_WINO_F4": "0"}),
            ("force_f4", {"E2V_CONV_ALGO": "3"})]
for name, env in VARIANTS:
    for k in ("E2V_CONV_ALGO", "E2V_WINO_F4", "E2V_WINO_F4_PAD", "E2V_WINO_MIN_C", "E2V_WINO_F4_MIN_C"):
        os.environ.pop(k, None)
    os.environ.update(env)
    from eeg2video_amd.pipeline import build_pipeline
    pipe = build_pipeline(UNetConfig(), VAEConfig(), device=0)
    eng = pipe.unet.engine
    out = eng.generate(lat, cond, unc, steps, 12.5, 0.0, decode=True, return_latents=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = eng.generate(lat, cond, unc, steps, 12.5, 0.0, decode=True, return_latents=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[name] = [o.float().cpu() for o in (out if isinstance(out, (tuple, list)) else [out])]
    print(name, f"{dt:.3f} s per pass of {B} clips", flush=True)
    del pipe, eng
    torch.cuda.empty_cache()
ref = res["direct"]
for name, _ in VARIANTS[1:]:
    for i, (a, b) in enumerate(zip(res[name], ref)):
        err = (a - b).abs().max().item()
        print(f"{name} vs direct, output {i} shape {tuple(a.shape)}: max abs {err:.3e}  / max ref {b.abs().max().item():.3e} = {err / b.abs().max().item():.3e}")
