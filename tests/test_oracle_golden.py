"""CPU: the oracle against the golden vectors captured from the reference itself
(tests/golden/make_golden.py) and against the closed-form known answers of SURVEY App. C."""
import os

import numpy as np
import pytest
import torch

from eeg2video_amd.weights import TINY_UNET, synth_state_dict, unet_param_spec
from oracle import DDIMOracle, unet3d_forward
from oracle import unet3d as O

TOL = dict(rtol=2e-5, atol=2e-5)   # fp32 CPU vs fp32 CPU, different op grouping only


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _sd(prefix, shapes, seed):
    spec = {prefix + k: v for k, v in shapes.items()}
    return {k[len(prefix):]: _t(v) for k, v in synth_state_dict(spec, seed=seed, mode="perturbed").items()}


@pytest.fixture(scope="module")
def t1(golden_dir):
    return np.load(os.path.join(golden_dir, "reference_t1_resnet.npz"))


@pytest.fixture(scope="module")
def t2(golden_dir):
    return np.load(os.path.join(golden_dir, "reference_t2_unet_tiny.npz"))


def _res_shapes(cin, cout, temb):
    s = {"norm1.weight": (cin,), "norm1.bias": (cin,), "conv1.weight": (cout, cin, 3, 3), "conv1.bias": (cout,),
         "time_emb_proj.weight": (cout, temb), "time_emb_proj.bias": (cout,),
         "norm2.weight": (cout,), "norm2.bias": (cout,), "conv2.weight": (cout, cout, 3, 3), "conv2.bias": (cout,)}
    if cin != cout:
        s["conv_shortcut.weight"] = (cout, cin, 1, 1)
        s["conv_shortcut.bias"] = (cout,)
    return s


def test_t1_inflated_conv(t1):
    sd = _sd("t1.conv.", {"weight": (64, 32, 3, 3), "bias": (64,)}, 11)
    y = O.inflated_conv3d(_t(t1["t1.x32"]), sd["weight"], sd["bias"])
    np.testing.assert_allclose(y.numpy(), t1["t1.conv.out"], **TOL)


@pytest.mark.parametrize("tag,cin,key", [("same", 64, "t1.x64"), ("proj", 32, "t1.x32")])
def test_t1_resnet_block(t1, tag, cin, key):
    sd = _sd(f"t1.res_{tag}.", _res_shapes(cin, 64, 128), 11)
    sd = {"r." + k: v for k, v in sd.items()}
    y = O.resnet_block3d(sd, "r", _t(t1[key]), _t(t1["t1.temb"]), groups=8, eps=1e-5)
    np.testing.assert_allclose(y.numpy(), t1[f"t1.res_{tag}.out"], **TOL)


def test_t1_downsample(t1):
    sd = _sd("t1.down.", {"conv.weight": (64, 64, 3, 3), "conv.bias": (64,)}, 11)
    sd = {"d." + k: v for k, v in sd.items()}
    y = O.downsample3d(sd, "d", _t(t1["t1.x64"]))
    np.testing.assert_allclose(y.numpy(), t1["t1.down.out"], **TOL)


@pytest.mark.parametrize("size,key", [(None, "out_x2"), ((3, 9, 16), "out_9x16"), ((3, 7, 11), "out_7x11")])
def test_t1_upsample(t1, size, key):
    sd = _sd("t1.up.", {"conv.weight": (64, 64, 3, 3), "conv.bias": (64,)}, 11)
    sd = {"u." + k: v for k, v in sd.items()}
    y = O.upsample3d(sd, "u", _t(t1["t1.x64"]), size)
    np.testing.assert_allclose(y.numpy(), t1[f"t1.up.{key}"], **TOL)


def test_t2_unet_forward_and_taps(t2):
    cfg = TINY_UNET
    sd = {k: _t(v) for k, v in synth_state_dict(unet_param_spec(cfg), seed=42, mode="perturbed").items()}
    x, cond = _t(t2["t2.unet.x"]), _t(t2["t2.unet.cond"])
    taps = {}
    y = unet3d_forward(sd, cfg, x, 501, cond, taps=taps)
    np.testing.assert_allclose(y.numpy(), t2["t2.unet.out_t501"], rtol=1e-4, atol=1e-4)
    for name in ("down0", "down1", "down2", "down3", "mid", "up0", "up1", "up2", "up3"):
        np.testing.assert_allclose(taps[name].numpy(), t2[f"t2.unet.tap.{name}"], rtol=1e-4, atol=1e-4,
                                   err_msg=name)
    y2 = unet3d_forward(sd, cfg, x, torch.tensor([751, 1]), cond)
    np.testing.assert_allclose(y2.numpy(), t2["t2.unet.out_t751_1"], rtol=1e-4, atol=1e-4)


def test_t2_transformer3d(t2):
    c, cross = 64, TINY_UNET.cross_attention_dim
    from eeg2video_amd.weights import _transformer3d
    spec = {}
    _transformer3d(spec, "t2.tr", c, cross)
    sd = {k[len("t2.tr."):]: _t(v) for k, v in synth_state_dict(spec, seed=13, mode="perturbed").items()}
    sd = {"a." + k: v for k, v in sd.items()}
    y = O.transformer3d(sd, "a", _t(t2["t2.tr.x"]), _t(t2["t2.tr.cond"]), heads=8, groups=32)
    np.testing.assert_allclose(y.numpy(), t2["t2.tr.out"], rtol=1e-4, atol=1e-4)


def test_t2_transformer3d_linear_projection_is_the_same_gemm():
    """Transformer3DModel(use_linear_projection=True) (attention.py:60-63,83-86,99-123: proj_in / proj_out as nn.Linear on the tokens)
    run by the REFERENCE itself (tests/golden/reference_t2_linear_projection.npz, make_golden.py --only-linear-projection): the oracle
    -- and the mirror's UNet3DConditionModel(use_linear_projection=True).load_state_dict -- treat the two [C, C] weights as 1x1 convs
    ([C, C, 1, 1]); on channel-last rows that is the same GEMM, and the reference's output says so."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_t2_linear_projection.npz"))
    sd = {}
    for k in g.files:
        if k.startswith("t2.trlin.w."):
            v = _t(g[k])
            name = k[len("t2.trlin.w."):]
            if name in ("proj_in.weight", "proj_out.weight"):
                assert v.ndim == 2
                v = v.reshape(v.shape[0], v.shape[1], 1, 1)
            sd["a." + name] = v
    y = O.transformer3d(sd, "a", _t(g["t2.trlin.x"]), _t(g["t2.trlin.cond"]), heads=8, groups=32)
    np.testing.assert_allclose(y.numpy(), g["t2.trlin.out"], rtol=1e-4, atol=1e-4)


# ---- closed-form known answers for the dependency-owned pieces ----
def test_ddim_timesteps_bit_exact():
    s = DDIMOracle()
    assert s.set_timesteps(4).tolist() == [751, 501, 251, 1]
    t50 = s.set_timesteps(50)
    assert t50.dtype == np.int64 and t50.tolist() == list(range(981, 0, -20))
    assert s.set_timesteps(100).tolist() == list(range(991, 0, -10))


def test_ddim_alpha_table():
    a = DDIMOracle().alphas_cumprod.numpy()
    known = {0: 0.99914998, 1: 0.99829602, 21: 0.98038065, 251: 0.67215115, 501: 0.27499884,
             751: 0.055718984, 961: 0.0072817220, 981: 0.0057754959, 999: 0.0046600951}
    for i, v in known.items():
        assert abs(a[i] - v) <= 2e-7 * max(1.0, abs(v)) + 1e-9, (i, a[i], v)


def test_ddim_step_matches_inversion_restatement():
    """tuneavideo/util.py:56-66 restates the same update (forward direction); stepping from t to
    t_prev with the model output held fixed and back must be the identity."""
    s = DDIMOracle()
    s.set_timesteps(50)
    x = torch.randn(2, 4, 3, 5, 6, generator=torch.Generator().manual_seed(0))
    eps = torch.randn(2, 4, 3, 5, 6, generator=torch.Generator().manual_seed(1))
    t = 501
    x_prev = s.step(eps, t, x)
    a_t, a_p = s.alphas_cumprod[t], s.alphas_cumprod[t - 20]
    # util.py:60-65 with timestep := t-20, next_timestep := t
    x0 = (x_prev - (1 - a_p) ** 0.5 * eps) / a_p ** 0.5
    x_back = a_t ** 0.5 * x0 + (1 - a_t) ** 0.5 * eps
    np.testing.assert_allclose(x_back.numpy(), x.numpy(), rtol=1e-5, atol=1e-5)


def test_sinusoid_known_values():
    e = O.timestep_sinusoid(torch.tensor([1]), 320)
    w = np.exp(-np.log(10000.0) * np.arange(160) / 160.0)
    assert abs(w[1] - 0.94406086) < 1e-7 and abs(w[159] - 1.0592537e-4) < 1e-10
    np.testing.assert_allclose(e[0, :160].numpy(), np.cos(w), rtol=1e-6, atol=1e-6)   # [cos, sin] (flip)
    np.testing.assert_allclose(e[0, 160:].numpy(), np.sin(w), rtol=1e-6, atol=1e-6)


# ---- SURVEY 8(f) ranks 1-2, pinned by running the reference's own CLIP and Diffusion (tier 1) ----
@pytest.fixture(scope="module")
def tx(golden_dir):
    return np.load(os.path.join(golden_dir, "reference_t1_extras.npz"))


def test_t1_dana_schedule_and_forward(tx):
    from oracle import dana_noise
    betas = torch.linspace(0.0001, 0.02, 500)
    ac = torch.cumprod(1.0 - betas, 0)
    np.testing.assert_array_equal(torch.sqrt(ac).numpy(), tx["x.dana.sqrt_alphas_cumprod"])
    np.testing.assert_array_equal(torch.sqrt(1 - ac).numpy(), tx["x.dana.sqrt_one_minus_alphas_cumprod"])
    out = dana_noise(_t(tx["x.dana.x0"]), _t(tx["x.dana.eps_div"]), _t(tx["x.dana.eps_same"]), _t(tx["x.dana.t"]), 0.3)
    ref = _t(tx["x.dana.out_beta0.3"]).permute(0, 2, 1, 3, 4)          # the caller's 'a b c d e -> a c b d e'
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=1e-6, atol=1e-6)


def test_t1_semantic_predictor_full_size(tx):
    """The reference's CLIP at its real size (0.89 G parameters, counter-RNG weights) against the oracle."""
    from eeg2video_amd.weights import SemanticConfig, semantic_param_spec
    from oracle import semantic_predictor
    sd = {k: _t(v) for k, v in synth_state_dict(semantic_param_spec(SemanticConfig(), 768), seed=44,
                                                 mode="reference_init").items()}
    with torch.no_grad():
        y = semantic_predictor(sd, _t(tx["x.clip.eeg"])).numpy()
    np.testing.assert_allclose(y[:, tx["x.clip.idx"]], tx["x.clip.out_sampled"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(np.sqrt((y.astype(np.float64) ** 2).sum(1)), tx["x.clip.out_l2"], rtol=1e-6)


# ---- DDIM inversion: the reference's own next_step / ddim_loop outputs ----------------------------
@pytest.fixture(scope="module")
def tinv(golden_dir):
    return np.load(os.path.join(golden_dir, "reference_t1_inversion.npz"))


def test_t1_inversion_next_step_bit_exact(tinv):
    """`next_step` (tuneavideo/util.py:56-66) run by the reference itself for n in {50, 4, 333, 20}: first / middle / last
    timestep (the first one exercises final_alpha_cumprod)."""
    from oracle import next_step
    s = DDIMOracle()
    assert np.array_equal(s.alphas_cumprod.numpy(), tinv["inv.alphas_cumprod"])
    eps, x = torch.from_numpy(tinv["inv.eps"]), torch.from_numpy(tinv["inv.x"])
    for n, t in tinv["inv.next_step.cases"]:
        s.set_timesteps(int(n))
        assert torch.equal(next_step(eps, int(t), x, s), torch.from_numpy(tinv[f"inv.next_step.n{n}.t{t}"])), (n, t)


def test_t1_inversion_loop(tinv):
    """`ddim_inversion` (util.py:74-101) run unmodified with a closed-form UNet stand-in: pins the ascending timestep order,
    the fp16 cast of the cond embeddings and the list it returns."""
    from oracle import ddim_loop
    s = DDIMOracle()
    n = int(tinv["inv.loop.n"][0])
    s.set_timesteps(n)
    cond16 = torch.from_numpy(tinv["inv.loop.cond"]).to(torch.float16)           # util.py:81

    def unet_fn(latents, t, cond):
        c = cond.float().mean(dim=(1, 2)).view(-1, 1, 1, 1, 1)
        return 0.3 * latents + 0.05 * torch.sin(latents * 3.0) + c + float(t) * 1e-4

    lat = ddim_loop(unet_fn, s, torch.from_numpy(tinv["inv.x"]), n, cond16)
    ref = tinv["inv.loop.latents"]
    assert len(lat) == n + 1 == ref.shape[0]
    for a, b in zip(lat, ref):
        assert torch.equal(a, torch.from_numpy(b))


# ---- PNDM / PLMS (dependency-owned, parity unpinned: closed forms and invariants only) ----------------
def test_pndm_oracle_closed_forms():
    from oracle import PNDMOracle
    s = PNDMOracle()
    assert s.set_timesteps(4).tolist() == [751, 501, 501, 251, 1]
    ts = s.set_timesteps(50)
    assert len(ts) == 51 and ts[:4].tolist() == [981, 961, 961, 941] and ts[-1] == 1
    # a constant model output is a fixed point of every multistep combination (coefficients sum to 1), so the whole PLMS
    # trajectory then equals repeated application of the transfer formula with that constant
    s.set_timesteps(6)
    g = torch.Generator().manual_seed(0)
    eps, x = torch.randn(2, 3, generator=g), torch.randn(2, 3, generator=g)
    y = x
    for t in s.timesteps:
        y = s.step(eps, int(t), y)
    z, ratio = x, 1000 // 6
    uniq = [int(t) for i, t in enumerate(s.timesteps) if i != 1]          # the repeated timestep re-does the first move
    for t in uniq:
        z = s._get_prev_sample(z, t, t - ratio, eps)
    assert torch.allclose(y, z, rtol=1e-5, atol=1e-6)
    # the first PLMS move is the eta = 0 DDIM move (formula (9) of PNDM reduces to it for a single model output)
    d = DDIMOracle()
    d.set_timesteps(6)
    s.set_timesteps(6)
    t0 = int(s.timesteps[0])
    assert torch.allclose(s.step(eps, t0, x), d.step(eps, t0, x), rtol=1e-5, atol=1e-6)
