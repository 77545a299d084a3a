import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from eeg2video_amd.pipeline import build_pipeline
from eeg2video_amd.weights import UNetConfig, VAEConfig, counter_normal, synth_state_dict, vae_param_spec
from oracle import vae_encode
vcfg = VAEConfig()
vsd = synth_state_dict(vae_param_spec(vcfg), seed=43, mode="reference_init")
pipe = build_pipeline(UNetConfig(), vcfg, device=0, vae_sd=vsd)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
img = t(counter_normal(77, "img", (1, 3, 288, 512))) * 0.5
mean_ref, logvar_ref = vae_encode({k: t(v) for k, v in vsd.items()}, vcfg, img)
post = pipe.vae.encode(img.cuda()).latent_dist
m, lv = post.mean.cpu(), post.logvar.cpu()
print("full-size VAE encode: mean err/scale %.2e  logvar err/scale %.2e" % ((m - mean_ref).abs().max() / mean_ref.abs().max(), (lv - logvar_ref).abs().max() / logvar_ref.abs().max()))
