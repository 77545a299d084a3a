import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from eeg2video_amd.pipeline import build_pipeline
from eeg2video_amd.weights import UNetConfig, VAEConfig, counter_normal
pipe = build_pipeline(UNetConfig(), VAEConfig(), device=0)
pipe.set_progress_bar_config(disable=True)
eng = pipe.unet.engine
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
for B in (1, 3):
    lat = torch.stack([t(counter_normal(1234 + k, "latent", (4, 6, 36, 64))) for k in range(B)]).cuda()
    cond = torch.stack([t(counter_normal(1235 + k, "cond", (77, 768))) for k in range(B)]).cuda()
    unc = t(counter_normal(1236, "uncond", (1, 77, 768))).cuda()
    fused, flat = eng.generate(lat, cond, unc, 3, 12.5, 0.0, decode=True, return_latents=True)
    # stepped: the public UNet entry point on the materialised pair + the fused update kernel
    ts = eng.ddim_timesteps(3)
    x = lat.clone()
    emb = torch.cat([unc.expand(B, -1, -1), cond])
    for tt in ts:
        eps = pipe.unet(torch.cat([x, x]), int(tt), encoder_hidden_states=emb).sample
        x = eng.ddim_cfg_step(eps[:B], eps[B:], x, 12.5, int(tt), int(tt) - 1000 // 3)
    print("B", B, "fused vs stepped latents bit-identical:", torch.equal(x, flat), " frames finite:", bool(torch.isfinite(fused).all()),
          " guidance off:", tuple(eng.generate(lat, cond, None, 2, 1.0, 0.0, decode=False, return_latents=True)[1].shape))
print("device bytes GiB", eng.device_bytes() / 2**30)
