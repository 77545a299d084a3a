"""CPU: closed-form anchors for the part of the oracle that no reference-held fixture can pin -- the arithmetic owned by the
absent dependency diffusers==0.11.1 (AutoencoderKL, CrossAttention._attention, GEGLU, Timesteps, DDIMScheduler.step).  Each
test checks the restatement against an independent statement of the published algorithm: a committed key list, torch's own
fused operators in fp64, fp64 closed forms, an algebraic identity with the reference-owned ``next_step``."""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from eeg2video_amd.weights import TINY_VAE, VAEConfig, counter_normal, synth_state_dict, vae_param_spec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_vae_key_set_and_shapes_match_the_sd_v1_4_layout():
    """248 tensors, 83 653 863 parameters: the list was written out block by block from the published SD-v1-4 vae/config.json
    layout (tests/golden/sd_v1_4_vae_keys.json), independently of eeg2video_amd.weights."""
    want = {k: tuple(v) for k, v in json.load(open(os.path.join(GOLDEN, "sd_v1_4_vae_keys.json"))).items()}
    got = dict(vae_param_spec(VAEConfig()))
    assert len(want) == 248 and sum(math.prod(v) for v in want.values()) == 83_653_863
    assert set(got) == set(want), set(got) ^ set(want)
    assert all(tuple(got[k]) == want[k] for k in want)


def test_vae_encode_decode_round_trip_shapes_at_full_size():
    """decode(mean(encode(x))) at 288x512: 8x down then 8x up, finite, and the F.pad (0,1,0,1) + stride-2 stack lands
    exactly on 36x64 (an off-by-one in any of the three downsamplers would show here)."""
    from oracle import vae_decode, vae_encode
    cfg = VAEConfig()
    sd = {k: _t(v) for k, v in synth_state_dict(vae_param_spec(cfg), seed=43, mode="reference_init").items()}
    img = _t(counter_normal(77, "img", (1, 3, 288, 512))) * 0.5
    with torch.no_grad():
        mean, logvar = vae_encode(sd, cfg, img)
        assert mean.shape == logvar.shape == (1, 4, 36, 64)
        assert torch.isfinite(mean).all() and float(logvar.min()) >= -30.0 and float(logvar.max()) <= 20.0
        out = vae_decode(sd, cfg, mean)
    assert out.shape == (1, 3, 288, 512) and torch.isfinite(out).all()


def test_vae_attention_block_matches_sdpa_in_fp64():
    """AttentionBlock = GroupNorm -> q, k, v -> softmax(q k^T / sqrt(C)) v -> proj + residual, one head."""
    from oracle.vae import _attention_block
    c, h, w = 32, 5, 7
    g = torch.Generator().manual_seed(1)
    sd = {}
    for n in ("query", "key", "value", "proj_attn"):
        sd[f"a.{n}.weight"] = torch.randn(c, c, generator=g, dtype=torch.float64) * 0.2
        sd[f"a.{n}.bias"] = torch.randn(c, generator=g, dtype=torch.float64) * 0.1
    sd["a.group_norm.weight"] = torch.randn(c, generator=g, dtype=torch.float64)
    sd["a.group_norm.bias"] = torch.randn(c, generator=g, dtype=torch.float64)
    x = torch.randn(2, c, h, w, generator=g, dtype=torch.float64)
    y = F.group_norm(x, 8, sd["a.group_norm.weight"], sd["a.group_norm.bias"], 1e-6).flatten(2).transpose(1, 2)
    q, k, v = (F.linear(y, sd[f"a.{n}.weight"], sd[f"a.{n}.bias"]) for n in ("query", "key", "value"))
    ref = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]       # default scale 1/sqrt(C)
    ref = F.linear(ref, sd["a.proj_attn.weight"], sd["a.proj_attn.bias"]).transpose(1, 2).reshape(2, c, h, w) + x
    got = _attention_block(sd, "a", x, 8, 1e-6)
    assert (got - ref).abs().max().item() < 5e-6         # the block's softmax is fp32 by construction (as the dependency's)


@pytest.mark.parametrize("d,nq,nk", [(40, 50, 77), (8, 33, 66), (160, 7, 14)])
def test_cross_attention_core_matches_sdpa_in_fp64(d, nq, nk):
    """CrossAttention._attention: baddbmm(alpha = d^-0.5) -> softmax -> bmm."""
    from oracle.unet3d import _attention
    g = torch.Generator().manual_seed(2)
    q, k, v = (torch.randn(6, n, d, generator=g, dtype=torch.float64) for n in (nq, nk, nk))
    ref = F.scaled_dot_product_attention(q, k, v)
    assert (_attention(q, k, v, d ** -0.5) - ref).abs().max().item() < 1e-12


def test_geglu_is_erf_gelu_and_the_kernel_erf_is_within_its_bound():
    """FeedForward(GEGLU): proj -> (value, gate) halves -> value * gelu(gate), exact (erf) gelu.  The HIP epilogue evaluates
    erf by Abramowitz-Stegun 7.1.26; its formula is restated here in fp64 and held to the published bound (|error| <=
    1.5e-7), which bounds the gelu error by 0.5 |x| 1.5e-7."""
    from oracle.unet3d import feed_forward
    g = torch.Generator().manual_seed(3)
    c = 16
    sd = {"ff.net.0.proj.weight": torch.randn(8 * c, c, generator=g, dtype=torch.float64), "ff.net.0.proj.bias": torch.randn(8 * c, generator=g, dtype=torch.float64),
          "ff.net.2.weight": torch.randn(c, 4 * c, generator=g, dtype=torch.float64), "ff.net.2.bias": torch.randn(c, generator=g, dtype=torch.float64)}
    x = torch.randn(3, 5, c, generator=g, dtype=torch.float64)
    h = F.linear(x, sd["ff.net.0.proj.weight"], sd["ff.net.0.proj.bias"])
    val, gate = h[..., :4 * c], h[..., 4 * c:]
    ref = F.linear(val * F.gelu(gate, approximate="none"), sd["ff.net.2.weight"], sd["ff.net.2.bias"])
    assert (feed_forward(sd, "ff", x) - ref).abs().max().item() < 1e-10
    z = np.linspace(0.0, 6.0, 200001)
    t = 1.0 / (1.0 + 0.3275911 * z)
    poly = ((((1.061405429 * t - 1.453152027) * t + 1.421413741) * t - 0.284496736) * t + 0.254829592) * t
    erf_as = 1.0 - poly * np.exp(-z * z)
    erf_exact = np.array([math.erf(v) for v in z[::100]])
    assert np.max(np.abs(erf_as[::100] - erf_exact)) <= 1.5e-7
    # bf16-activation mode: the gate as a logistic of an odd cubic (igemm_epi.h: gelu_bf16_grade), restated in fp64: within 2.8e-4 of
    # x Phi(x) on the whole line (1/14 of the rounding of a bf16 value near 1), exact limits 0 and x at -inf / +inf
    xs = np.linspace(-40.0, 40.0, 800001)
    fast = xs / (1.0 + np.exp(-(1.60031415 * xs + 0.06940179 * xs ** 3)))
    exact = 0.5 * xs * (1.0 + np.array([math.erf(v / math.sqrt(2.0)) for v in xs]))
    assert np.max(np.abs(fast - exact)) <= 2.8e-4
    assert abs(fast[0]) < 1e-30 and fast[-1] == xs[-1]


@pytest.mark.parametrize("t", [1, 501, 981])
def test_timestep_sinusoid_matches_the_fp64_closed_form(t):
    """get_timestep_embedding(flip_sin_to_cos=True, freq_shift=0), dim 320: emb[i] = cos(t w_i), emb[160 + i] = sin(t w_i),
    w_i = 10000^(-i / 160)."""
    from oracle.unet3d import timestep_sinusoid
    got = timestep_sinusoid(torch.tensor([t]), 320, True, 0)[0].double()
    i = torch.arange(160, dtype=torch.float64)
    w = torch.exp(-math.log(10000.0) * i / 160.0)
    ref = torch.cat([torch.cos(t * w), torch.sin(t * w)])
    # fp32 evaluation of t * w (t up to 981, w up to 1): absolute argument error ~ 981 * 6e-8
    assert (got - ref).abs().max().item() < 2e-4
    assert abs(got[0].item() - math.cos(t)) < 1e-4 and abs(got[160].item() - math.sin(t)) < 1e-4


def test_ddim_step_then_next_step_is_the_identity_for_every_golden_case():
    """DDIMScheduler.step (dependency-owned, restated) followed by the reference-owned next_step (tuneavideo/util.py:56-66,
    pinned bit-exactly by tests/golden/reference_t1_inversion.npz) with the same epsilon returns the sample it started from:
    x -> x_{t - T/n} -> x.  Run for every (n, t) the golden holds, in fp64 tables to separate algebra from rounding."""
    from oracle import DDIMOracle, next_step
    z = np.load(os.path.join(GOLDEN, "reference_t1_inversion.npz"))
    cases = sorted({(int(k.split(".")[2][1:]), int(k.split(".")[3][1:])) for k in z.files if k.startswith("inv.next_step.n")})
    assert len(cases) >= 12
    g = torch.Generator().manual_seed(5)
    for n, t in cases:
        s = DDIMOracle()
        s.alphas_cumprod = s.alphas_cumprod.double()
        s.final_alpha_cumprod = s.alphas_cumprod[0]
        s.set_timesteps(n)
        x = torch.randn(2, 4, 3, 5, 6, generator=g, dtype=torch.float64)
        eps = torch.randn(2, 4, 3, 5, 6, generator=g, dtype=torch.float64)
        back = next_step(eps, t, s.step(eps, t, x), s)
        assert (back - x).abs().max().item() < 1e-10, (n, t)


def test_generate_trace_holds_unguided_eps_and_block_taps():
    """oracle.generate(trace=...): per step the guided eps, the two unguided halves (eps = eps_u + g (eps_c - eps_u), checked) and the
    latents; with trace["taps_step"] = k also the block outputs of step k's UNet forward (the keys the HIP side's
    e2v_op_unet_forward_taps returns, shapes of a [2 B, ...] guidance pair) -- what the block-granularity GPU tests compare against."""
    from eeg2video_amd.weights import TINY_UNET, TINY_VAE, counter_normal, synth_state_dict, unet_param_spec, vae_param_spec
    from oracle import generate
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    usd = {k: t(v) for k, v in synth_state_dict(unet_param_spec(TINY_UNET), seed=42, mode="perturbed").items()}
    vsd = {k: t(v) for k, v in synth_state_dict(vae_param_spec(TINY_VAE), seed=43, mode="perturbed").items()}
    lat = t(counter_normal(1, "lat", (1, 4, 3, 8, 12)))
    cond = t(counter_normal(2, "cond", (1, 7, TINY_UNET.cross_attention_dim)))
    unc = t(counter_normal(3, "unc", (1, 7, TINY_UNET.cross_attention_dim)))
    trace = {"taps_step": 1}
    with torch.no_grad():
        generate(usd, TINY_UNET, vsd, TINY_VAE, lat, cond, unc, 2, 12.5, trace=trace, decode=False)
    assert len(trace["eps"]) == len(trace["eps_u"]) == len(trace["eps_c"]) == len(trace["latents"]) == 2
    for e, u, c in zip(trace["eps"], trace["eps_u"], trace["eps_c"]):
        assert torch.equal(e, u + 12.5 * (c - u))
    taps = trace["taps"]
    assert tuple(taps) == ("emb", "down0", "down1", "down2", "down3", "mid", "up0", "up1", "up2", "up3")
    boc = TINY_UNET.block_out_channels
    assert taps["emb"].shape == (2, 4 * boc[0]) and taps["down0"].shape[:3] == (2, boc[0], 3) and taps["up3"].shape == (2, boc[0], 3, 8, 12)
