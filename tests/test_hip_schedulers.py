"""GPU: the scheduler mirrors the pipeline's constructor accepts beside DDIM / PNDM (pipeline_tuneeeg2video.py:48-55) --
Euler, Euler-ancestral, LMS, DPM-Solver++ (2M) -- and the stochastic DDIM update (eta > 0): host-side tables and bookkeeping,
device arithmetic through e2v_lincomb, against oracle/schedulers.py per step on a random trajectory and through the whole
pipeline (fractional timesteps reach the UNet through e2v_unet_forward_ft)."""
import numpy as np
import pytest
import torch

from eeg2video_amd.weights import (TINY_UNET, TINY_VAE, counter_normal, synth_state_dict, unet_param_spec,
                                   vae_param_spec)

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.fixture(scope="module")
def tiny():
    from eeg2video_amd.pipeline import build_pipeline
    usd = synth_state_dict(unet_param_spec(TINY_UNET), seed=42, mode="perturbed")
    vsd = synth_state_dict(vae_param_spec(TINY_VAE), seed=43, mode="perturbed")
    pipe = build_pipeline(TINY_UNET, TINY_VAE, device=0, unet_sd=usd, vae_sd=vsd)
    pipe.set_progress_bar_config(disable=True)
    return pipe, {k: _t(v) for k, v in usd.items()}, {k: _t(v) for k, v in vsd.items()}


def _pairs():
    from eeg2video_amd import scheduler as S
    import oracle as O
    return {
        "euler": (S.EulerDiscreteScheduler, O.EulerOracle, 3e-6),
        "euler_a": (S.EulerAncestralDiscreteScheduler, O.EulerAncestralOracle, 3e-6),
        "lms": (S.LMSDiscreteScheduler, O.LMSOracle, 3e-4),          # the mirror integrates numerically (epsrel 1e-4), the oracle exactly
        "dpmpp": (S.DPMSolverMultistepScheduler, O.DPMSolverPPOracle, 3e-6),
    }


@pytest.mark.parametrize("name", ["euler", "euler_a", "lms", "dpmpp"])
@pytest.mark.parametrize("n", [7, 20])
def test_scheduler_steps_vs_oracle(tiny, name, n):
    eng = tiny[0].unet.engine
    mirror_cls, oracle_cls, tol = _pairs()[name]
    so, sm = oracle_cls(), mirror_cls(engine=eng)
    so.set_timesteps(n)
    sm.set_timesteps(n)
    assert np.array_equal(np.asarray(sm.timesteps), np.asarray(so.timesteps))
    assert abs(float(sm.init_noise_sigma) - float(so.init_noise_sigma)) < 1e-6 * float(so.init_noise_sigma)
    shape = (2, 4, 3, 5, 6)
    xo = _t(counter_normal(41, "x", shape)) * float(so.init_noise_sigma)
    xm = xo.cuda()
    for i, t in enumerate(so.timesteps):
        t = int(t) if float(t).is_integer() else float(t)
        eps = _t(counter_normal(42 + i, "e", shape))
        z = _t(counter_normal(900 + i, "z", shape))
        assert rel_err(sm.scale_model_input(xm, t), so.scale_model_input(xo, t)) < 2e-6, i
        xo = so.step(eps, t, xo, noise=z)
        kw = {"noise": z.cuda()} if name == "euler_a" else {}
        xm = sm.step(eps.cuda(), t, xm, **kw).prev_sample
        assert rel_err(xm, xo) < tol, (i, rel_err(xm, xo))


def test_ddim_eta_step_vs_oracle(tiny):
    from eeg2video_amd.scheduler import DDIMScheduler
    from oracle import DDIMOracle
    eng = tiny[0].unet.engine
    so, sm = DDIMOracle(), DDIMScheduler(engine=eng)
    so.set_timesteps(10)
    sm.set_timesteps(10)
    shape = (2, 4, 3, 5, 6)
    xo = _t(counter_normal(41, "x", shape))
    xm = xo.cuda()
    for i, t in enumerate(so.timesteps):
        eps, z = _t(counter_normal(42 + i, "e", shape)), _t(counter_normal(900 + i, "z", shape))
        xo = so.step(eps, int(t), xo, eta=0.7, noise=z)
        xm = sm.step(eps.cuda(), int(t), xm, eta=0.7, variance_noise=z.cuda()).prev_sample
        assert rel_err(xm, xo) < 3e-6, i
    # the generator path draws the same numbers as torch.randn with that generator
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    eps = _t(counter_normal(50, "e", shape))
    a = sm.step(eps.cuda(), 501, xm, eta=1.0, generator=g1).prev_sample
    z = torch.randn(shape, generator=g2)
    so.set_timesteps(10)
    assert rel_err(a, so.step(eps, 501, xm.cpu(), eta=1.0, noise=z)) < 3e-6


@pytest.mark.parametrize("name", ["euler", "euler_a", "lms", "dpmpp", "ddim_eta"])
def test_pipeline_with_each_scheduler_vs_oracle(tiny, name):
    """TuneAVideoPipeline.__call__ with the scheduler swapped in: 4 inference steps, guidance on; frames within 1e-3."""
    from eeg2video_amd.scheduler import DDIMScheduler
    from oracle import DDIMOracle, generate
    pipe, usd, vsd = tiny
    eng = pipe.unet.engine
    b, f, tok, d = 1, 3, 77, TINY_UNET.cross_attention_dim
    shape = (b, 4, f, 4, 6)
    lat = _t(counter_normal(70, "lat", shape))
    eeg = _t(counter_normal(71, "eeg", (b, tok * d)))
    neg = _t(counter_normal(72, "neg", (1, tok, d)))
    n, eta = 4, 0.0
    if name == "ddim_eta":
        mirror, orc, eta = DDIMScheduler(engine=eng), DDIMOracle(), 0.6
    else:
        mirror_cls, oracle_cls, _ = _pairs()[name]
        mirror, orc = mirror_cls(engine=eng), oracle_cls()
    g = torch.Generator().manual_seed(11)
    g2 = torch.Generator().manual_seed(11)
    noises = [torch.randn(shape, generator=g2) for _ in range(n)] if name in ("euler_a", "ddim_eta") else None
    ref = generate(usd, TINY_UNET, vsd, TINY_VAE, lat, eeg.reshape(b, tok, d), neg, n, 7.5, eta=eta, scheduler=orc, noises=noises)
    old = pipe.scheduler
    try:
        pipe.scheduler = mirror
        out = pipe(None, eeg, video_length=f, height=32, width=48, num_inference_steps=n, guidance_scale=7.5,
                   negative_prompt=neg, latents=lat.cuda(), eta=eta, generator=g).videos
    finally:
        pipe.scheduler = old
    assert out.shape == ref.shape and torch.isfinite(out).all()
    assert (out - ref).abs().max().item() < (2e-3 if name == "lms" else 1e-3), (out - ref).abs().max().item()
