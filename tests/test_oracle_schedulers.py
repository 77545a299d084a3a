"""CPU: closed-form anchors for the scheduler oracles of oracle/schedulers.py (dependency-owned algorithms, no reference
fixtures exist): fp64 tables, algebraic identities between schemes, and the exact-denoiser trajectory every consistent
scheme must reproduce."""
import math

import numpy as np
import pytest
import torch

from oracle import DDIMOracle, DPMSolverPPOracle, EulerAncestralOracle, EulerOracle, LMSOracle, lms_coefficient_exact


def _abar64():
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    return np.cumprod(1.0 - betas)


@pytest.mark.parametrize("n", [4, 20, 50])
def test_sigma_tables_and_timesteps(n):
    ab = _abar64()
    sig = np.sqrt((1 - ab) / ab)
    for cls in (EulerOracle, EulerAncestralOracle, LMSOracle):
        s = cls()
        ts = s.set_timesteps(n)
        assert np.allclose(ts, np.linspace(0, 999, n)[::-1]) and ts[0] == 999.0 and ts[-1] == 0.0
        want = np.interp(ts, np.arange(1000), sig)
        assert np.allclose(s.sigmas[:-1].numpy(), want, rtol=2e-5) and float(s.sigmas[-1]) == 0.0
        assert abs(s.init_noise_sigma - sig[-1]) < 2e-4 * sig[-1]                   # 14.6146 for Stable Diffusion
    d = DPMSolverPPOracle()
    assert d.set_timesteps(4).tolist() == [999, 749, 500, 250]
    assert np.array_equal(d.set_timesteps(n), np.linspace(0, 999, n + 1).round()[::-1][:-1].astype(np.int64))


def test_lms_coefficients():
    s = LMSOracle()
    s.set_timesteps(20)
    for t in (0, 3, 10, 18):
        for order in range(1, min(t + 1, 4) + 1):
            c = [lms_coefficient_exact(s.sigmas, order, t, j) for j in range(order)]
            # the Lagrange basis sums to one: the coefficients sum to the Euler step; order 1 IS the Euler step
            assert abs(sum(c) - float(s.sigmas[t + 1] - s.sigmas[t])) < 1e-5 * float(s.sigmas[t])
            # the dependency's own numerical integral (scipy quad, epsrel 1e-4) agrees to its tolerance
            from eeg2video_amd.scheduler import LMSDiscreteScheduler
            m = LMSDiscreteScheduler()
            m.set_timesteps(20)
            for j in range(order):
                q = m.get_lms_coefficient(order, t, j)
                assert abs(q - c[j]) <= 2e-4 * max(abs(cc) for cc in c) + 1e-7


def test_euler_ancestral_variance_split():
    s = EulerAncestralOracle()
    s.set_timesteps(10)
    for i in range(10):
        sf, st = float(s.sigmas[i]), float(s.sigmas[i + 1])
        up = (st ** 2 * (sf ** 2 - st ** 2) / sf ** 2) ** 0.5
        down = (st ** 2 - up ** 2) ** 0.5
        assert abs(up ** 2 + down ** 2 - st ** 2) < 1e-6 * max(st ** 2, 1e-6)
    # zero noise and an exact denoiser: the deterministic part walks to sigma_down, i.e. x0 + sigma_down n
    g = torch.Generator().manual_seed(0)
    x0, n = torch.randn(2, 3, generator=g, dtype=torch.float64), torch.randn(2, 3, generator=g, dtype=torch.float64)
    i = 4
    sf, st = float(s.sigmas[i]), float(s.sigmas[i + 1])
    up = (st ** 2 * (sf ** 2 - st ** 2) / sf ** 2) ** 0.5
    down = (st ** 2 - up ** 2) ** 0.5
    out = s.step(n, s.timesteps[i], x0 + sf * n, noise=torch.zeros_like(n))
    assert (out - (x0 + down * n)).abs().max().item() < 1e-5


@pytest.mark.parametrize("cls", [EulerOracle, LMSOracle])
def test_sigma_space_schemes_follow_the_exact_denoiser_trajectory(cls):
    """model = (x - x0) / sigma for a fixed x0: x_i = x0 + sigma_i n at every step, x0 at the end (sigma = 0)."""
    s = cls()
    ts = s.set_timesteps(12)
    g = torch.Generator().manual_seed(1)
    x0, n = torch.randn(2, 5, generator=g, dtype=torch.float64), torch.randn(2, 5, generator=g, dtype=torch.float64)
    x = x0 + float(s.sigmas[0]) * n
    for i, t in enumerate(ts):
        x = s.step(n, t, x)
        assert (x - (x0 + float(s.sigmas[i + 1]) * n)).abs().max().item() < 2e-5 * (1 + float(s.sigmas[i + 1]))
    assert (x - x0).abs().max().item() < 1e-5


def test_dpm_solver_first_order_is_the_ddim_update_and_second_order_keeps_the_exact_trajectory():
    ab = torch.from_numpy(_abar64())
    d = DPMSolverPPOracle()
    for tab in ("alphas_cumprod", "alpha_t", "sigma_t", "lambda_t"):
        setattr(d, tab, getattr(d, tab).double())
    d.alpha_t, d.sigma_t = torch.sqrt(ab), torch.sqrt(1 - ab)
    d.lambda_t = torch.log(d.alpha_t) - torch.log(d.sigma_t)
    ts = d.set_timesteps(10)
    g = torch.Generator().manual_seed(2)
    x, eps = torch.randn(2, 7, generator=g, dtype=torch.float64), torch.randn(2, 7, generator=g, dtype=torch.float64)
    t, prev = int(ts[0]), int(ts[1])
    out = d.step(eps, t, x)                                           # first step: first-order update
    x0 = (x - (1 - ab[t]) ** 0.5 * eps) / ab[t] ** 0.5
    ddim = ab[prev] ** 0.5 * x0 + (1 - ab[prev]) ** 0.5 * eps          # deterministic DDIM from t to prev
    assert (out - ddim).abs().max().item() < 1e-10
    # exact data prediction x0*: x_t = alpha_t x0* + sigma_t n at every step, whatever the order
    d.set_timesteps(10)
    x0s, n = torch.randn(2, 7, generator=g, dtype=torch.float64), torch.randn(2, 7, generator=g, dtype=torch.float64)
    x = d.alpha_t[ts[0]] * x0s + d.sigma_t[ts[0]] * n
    for i, t in enumerate(ts):
        t = int(t)
        eps_exact = (x - d.alpha_t[t] * x0s) / d.sigma_t[t]
        x = d.step(eps_exact, t, x)
        prev = int(ts[i + 1]) if i + 1 < len(ts) else 0
        assert (x - (d.alpha_t[prev] * x0s + d.sigma_t[prev] * n)).abs().max().item() < 1e-9


def test_ddim_eta_variance_and_reduction_to_the_deterministic_step():
    s = DDIMOracle()
    s.set_timesteps(20)
    g = torch.Generator().manual_seed(3)
    x, eps, z = (torch.randn(2, 4, 3, 5, 6, generator=g) for _ in range(3))
    t = 501
    assert torch.equal(s.step(eps, t, x, eta=0.0), s.step(eps, t, x))
    a_t, a_p = float(s.alphas_cumprod[t]), float(s.alphas_cumprod[t - 50])
    std = 1.0 * math.sqrt((1 - a_p) / (1 - a_t) * (1 - a_t / a_p))
    full = s.step(eps, t, x, eta=1.0, noise=z)
    x0 = (x - math.sqrt(1 - a_t) * eps) / math.sqrt(a_t)
    want = math.sqrt(a_p) * x0 + math.sqrt(1 - a_p - std ** 2) * eps + std * z
    assert (full - want).abs().max().item() < 1e-5


def test_scheduler_registry_builds_each_type_from_a_config():
    from eeg2video_amd.scheduler import SCHEDULERS, scheduler_from_config
    for name, cls in SCHEDULERS.items():
        cfg = {"_class_name": name, "_diffusers_version": "0.11.1", "beta_start": 0.00085, "beta_end": 0.012,
               "beta_schedule": "scaled_linear", "num_train_timesteps": 1000, "trained_betas": None}
        if name in ("DDIMScheduler", "PNDMScheduler"):
            cfg.update(set_alpha_to_one=False, steps_offset=1)
        if name == "PNDMScheduler":
            cfg.update(skip_prk_steps=True)
        s = scheduler_from_config(cfg)
        assert isinstance(s, cls) and s.config.num_train_timesteps == 1000
    with pytest.raises(ValueError):
        scheduler_from_config({"_class_name": "KarrasVeScheduler"})
