"""One bf16 sparse-causal attention launch at the level-0 shape (B = 32: 64 samples x 6 frames x 8 heads, 2304 queries, d = 40), for the
counter passes of tools/pmc_micro.sh (MICRO=tools/attn_pmc.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg2video_amd.engine import Engine
from eeg2video_amd.weights import TINY_UNET, TINY_VAE
eng = Engine(TINY_UNET, TINY_VAE, 0)
eng.set_compute_dtype(os.environ.get("DTYPE", "bf16"))
n, f, nq, d, heads = int(os.environ.get("SAMPLES", "64")), 6, 2304, 40, 8
c = heads * d
qkv = torch.randn(n * f * nq, 3 * c, device="cuda")
for _ in range(int(os.environ.get("REPS", "2"))):
    eng.op_attention(qkv[:, :c], qkv[:, c:2*c], qkv[:, 2*c:], n=n, F=f, heads=heads, D=d, Nq=nq, Nk=nq, mode=0, scale=d ** -0.5)
torch.cuda.synchronize()
