import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/eeg2video_amd") else os.environ["GRAFT_REPO_ROOT"])
import torch, torch.nn.functional as F
from eeg2video_amd.engine import Engine
from eeg2video_amd.weights import TINY_UNET, TINY_VAE
eng = Engine(TINY_UNET, TINY_VAE, 0)   # E2V_F32X3=1 in env
torch.manual_seed(0)
eng.profile_begin()
for m, k, n in [(240, 1280, 1280), (1000, 320, 960), (77, 64, 128), (5, 320, 1280), (4096, 640, 640), (130, 40, 72)]:
    x, w, b, r = torch.randn(m, k), torch.randn(n, k) * 0.05, torch.randn(n), torch.randn(m, n)
    ref = (x.double() @ w.double().T + b.double() + r.double())
    y = eng.op_linear(x.cuda(), w.cuda(), b.cuda(), r.cuda()).cpu().double()
    y32 = (F.linear(x, w, b) + r).double()
    s = ref.abs().max()
    print(m, k, n, "x3 err/scale %.2e   torch-fp32 err/scale %.2e" % ((y - ref).abs().max() / s, (y32 - ref).abs().max() / s))
x, w, b = torch.randn(864, 320), torch.randn(2560, 320) * 0.1, torch.randn(2560)
h, g = F.linear(x.double(), w.double(), b.double()).chunk(2, dim=-1)
ref = h * F.gelu(g)
y = eng.op_linear(x.cuda(), w.cuda(), b.cuda(), geglu=True).cpu().double()
print("geglu err/scale %.2e" % ((y - ref).abs().max() / ref.abs().max()))
print("kernel classes used:", sorted(eng.profile_end().keys()))
