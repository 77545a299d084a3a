"""GPU: the SD-v1-4 sized path (BASELINE.json shapes) -- one UNet sample and one VAE frame against the CPU
oracle, plus size-independent properties at the full benchmark shapes."""
import numpy as np
import pytest
import torch

from eeg2video_amd.weights import UNetConfig, VAEConfig, counter_normal, synth_state_dict, unet_param_spec, vae_param_spec

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.fixture(scope="module")
def full():
    from eeg2video_amd.pipeline import build_pipeline
    ucfg, vcfg = UNetConfig(), VAEConfig()
    usd = synth_state_dict(unet_param_spec(ucfg), seed=42, mode="reference_init")
    vsd = synth_state_dict(vae_param_spec(vcfg), seed=43, mode="reference_init")
    pipe = build_pipeline(ucfg, vcfg, device=0, unet_sd=usd, vae_sd=vsd)
    pipe.set_progress_bar_config(disable=True)
    return pipe, usd, vsd


def test_full_unet_sample_vs_oracle(full):
    """[1,4,6,36,64] latent + [1,77,768] cond, t = 501: 2.96 TFLOP on the CPU oracle (~15-25 s)."""
    from oracle import unet3d_forward
    pipe, usd, _ = full
    cfg = UNetConfig()
    x = _t(counter_normal(1234, "latent", (1, 4, 6, 36, 64)))
    cond = _t(counter_normal(1235, "cond", (1, 77, 768)))
    y = pipe.unet(x.cuda(), 501, cond.cuda()).sample
    ref = unet3d_forward({k: _t(v) for k, v in usd.items()}, cfg, x, 501, cond)
    assert y.shape == (1, 4, 6, 36, 64)
    err = rel_err(y, ref)
    print(f"full UNet sample: max-abs / max-ref = {err:.3e}")
    assert err < 1e-3            # north_star tolerance


def test_full_vae_frame_vs_oracle(full):
    from oracle import vae_decode
    pipe, _, vsd = full
    z = _t(counter_normal(77, "z", (1, 4, 36, 64)))
    y = pipe.vae.decode(z.cuda()).sample
    ref = vae_decode({k: _t(v) for k, v in vsd.items()}, VAEConfig(), z)
    err = rel_err(y, ref)
    print(f"full VAE frame: max-abs / max-ref = {err:.3e}")
    assert y.shape == (1, 3, 288, 512) and err < 1e-3


def test_full_batch_properties(full):
    """At the benchmark shape (CFG pair of 2 clips = 4 UNet samples): batch entries are independent
    (bit-exact against the single-sample call) and the fused loop equals the stepped loop."""
    pipe = full[0]
    eng = pipe.unet.engine
    x = _t(counter_normal(1234, "latent", (2, 4, 6, 36, 64))).cuda()
    cond = _t(counter_normal(1235, "cond", (2, 77, 768))).cuda()
    unc = _t(counter_normal(1236, "uncond", (1, 77, 768))).cuda()
    both = pipe.unet(x, 981, cond).sample
    one = pipe.unet(x[1:], 981, cond[1:]).sample
    assert torch.equal(both[1:], one)
    lat2 = eng.generate(x, cond, unc, 2, 12.5, 0.0, decode=False, return_latents=True)[1]
    ts = eng.ddim_timesteps(2)
    cur = x
    emb = torch.cat([unc.expand(2, -1, -1), cond])
    for t in ts:
        eps = pipe.unet(torch.cat([cur, cur]), int(t), emb).sample
        cur = eng.ddim_cfg_step(eps[:2], eps[2:], cur, 12.5, int(t), int(t) - 500)
    assert torch.equal(cur, lat2)
    vid = eng.vae_decode(lat2, postprocess=True)
    assert vid.shape == (2, 3, 6, 288, 512) and float(vid.min()) >= 0.0 and float(vid.max()) <= 1.0
    assert torch.isfinite(vid).all()
