import json, os, sys
os.environ["E2V_PROFILE_DETAIL"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eeg2video_amd.pipeline import build_pipeline
from eeg2video_amd.weights import UNetConfig, VAEConfig, counter_normal
pipe = build_pipeline(UNetConfig(), VAEConfig(), device=0)
eng = pipe.unet.engine
if "DTYPE" in os.environ:
    eng.set_compute_dtype(os.environ["DTYPE"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
lat = torch.stack([t(counter_normal(1234 + k, "latent", (4, 6, 36, 64))) for k in range(B)]).cuda()
cond = torch.stack([t(counter_normal(1235 + k, "cond", (77, 768))) for k in range(B)]).cuda()
unc = t(counter_normal(1236, "uncond", (1, 77, 768))).cuda()
DEC = os.environ.get("DECODE", "1") == "1"
eng.generate(lat, cond, unc, 1, 12.5, 0.0, decode=DEC)
torch.cuda.synchronize()
eng.profile_begin()
eng.generate(lat, cond, unc, 1, 12.5, 0.0, decode=DEC)
tab = eng.profile_end()
tot = sum(v["ms"] for v in tab.values())
rows = sorted(tab.items(), key=lambda kv: -kv[1]["ms"])
for k, v in rows[:60]:
    print(f"{v['ms']:9.2f} ms {100*v['ms']/tot:5.1f}%  n={v['launches']:4d} {v['flops']/v['ms']/1e9:7.1f} TF {v['bytes']/v['ms']/1e6:7.0f} GB/s  {k}")
print("total ms", tot)
json.dump(tab, open(os.environ.get("OUT", "gpurun_out/shape_profile.json"), "w"), indent=1)
