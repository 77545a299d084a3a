"""f32x3 attention against the fp32-MFMA attention and an fp64 reference (cross-attention form), plus timing."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eeg2video_amd.engine import Engine
from eeg2video_amd.weights import TINY_UNET, TINY_VAE
eng = Engine(TINY_UNET, TINY_VAE, 0)
torch.manual_seed(0)
def run(mode_name, **kw):
    eng.set_compute_dtype(mode_name)
    return eng.op_attention(**kw)
for name, n, f, nq, nk, d, mode in [("sparse d40", 2, 6, 300, 300, 40, 0), ("sparse d80", 2, 3, 576, 576, 80, 0), ("sparse d160", 1, 6, 144, 144, 160, 0),
                                    ("cross d40 77 keys", 2, 3, 200, 77, 40, 1), ("sparse d8 ragged", 1, 3, 45, 45, 8, 0)]:
    heads = 8; c = heads * d
    q = torch.randn(n * f * nq, c, device="cuda")
    kvrows = n * f * nk if mode == 0 else n * nk
    k = torch.randn(kvrows, c, device="cuda"); v = torch.randn(kvrows, c, device="cuda")
    kw = dict(q=q, k=k, v=v, n=n, F=f, heads=heads, D=d, Nq=nq, Nk=nk, mode=mode, scale=d ** -0.5)
    y32 = run("fp32", **kw); y3 = run("f32x3", **kw)
    err = (y3 - y32).abs().max().item() / y32.abs().max().item()
    msg = f"{name}: x3 vs fp32-MFMA {err:.2e}"
    if mode == 1:
        qd, kd, vd = (t.double().cpu() for t in (q, k, v))
        ref = torch.empty_like(qd)
        for s_ in range(n):
            for h_ in range(heads):
                sl = slice(h_ * d, (h_ + 1) * d)
                qq = qd[s_ * f * nq:(s_ + 1) * f * nq, sl]; kk = kd[s_ * nk:(s_ + 1) * nk, sl]; vv = vd[s_ * nk:(s_ + 1) * nk, sl]
                ref[s_ * f * nq:(s_ + 1) * f * nq, sl] = torch.softmax(qq @ kk.T * d ** -0.5, -1) @ vv
        sc = ref.abs().max().item()
        msg += f"   vs fp64: x3 {(y3.double().cpu() - ref).abs().max().item() / sc:.2e}  fp32 {(y32.double().cpu() - ref).abs().max().item() / sc:.2e}"
    print(msg)
for name, n, f, nq, d in [("L0 d40", 16, 6, 2304, 40), ("L1 d80", 16, 6, 576, 80), ("L2 d160", 16, 6, 144, 160)]:
    heads = 8; c = heads * d
    qkv = torch.randn(n * f * nq, 3 * c, device="cuda")
    for mode_name in ("fp32", "f32x3"):
        eng.set_compute_dtype(mode_name)
        fn = lambda: eng.op_attention(qkv[:, :c], qkv[:, c:2*c], qkv[:, 2*c:], n=n, F=f, heads=heads, D=d, Nq=nq, Nk=nq, mode=0, scale=d ** -0.5)
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"{name} {mode_name}: {min(ts)*1e3:.3f} ms")
