import os, sys
os.environ["E2V_PROFILE_DETAIL"] = "1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from eeg2video_amd.pipeline import build_pipeline
from eeg2video_amd.weights import UNetConfig, VAEConfig, counter_normal
pipe = build_pipeline(UNetConfig(), VAEConfig(), device=0)
eng = pipe.unet.engine
z = torch.randn(8, 4, 6, 36, 64, device="cuda")
eng.vae_decode(z); torch.cuda.synchronize()
eng.profile_begin(); eng.vae_decode(z); tab = eng.profile_end()
tot = sum(v["ms"] for v in tab.values())
for k, v in sorted(tab.items(), key=lambda kv: -kv[1]["ms"])[:28]:
    print(f"{v['ms']:9.2f} ms {100*v['ms']/tot:5.1f}%  n={v['launches']:4d} {v['flops']/v['ms']/1e9:7.1f} TF {v['bytes']/v['ms']/1e6:7.0f} GB/s  {k}")
print("total ms", tot)
