"""Oracle: the remaining scheduler types ``TuneAVideoPipeline.__init__`` accepts
(``EEG2Video/pipelines/pipeline_tuneeeg2video.py:48-55``): ``EulerDiscreteScheduler``, ``EulerAncestralDiscreteScheduler``,
``LMSDiscreteScheduler``, ``DPMSolverMultistepScheduler`` (DPM-Solver++ 2M, the Stable-Diffusion configuration).  torch CPU, fp32 tables as the
dependency builds them.  (The stochastic branch of DDIM, ``eta > 0``, lives in ``oracle/ddim.py``.)

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  All four belong to the absent dependency ``diffusers==0.11.1``
(``requirements.txt:4``); restated from their published algorithms (k-diffusion's ``sample_euler`` / ``sample_euler_ancestral`` /
``sample_lms`` in sigma space; DPM-Solver++ formula (2M) with the midpoint rule and a first-order last step below 15 steps) --
PARITY UNPINNED by the reference (it holds no fixtures for them), anchored on closed forms in ``tests/test_oracle_schedulers.py``:
sigma tables, the first-order DPM-Solver++ update = the deterministic DDIM update, the order-1 LMS coefficient = the Euler step,
``sigma_up^2 + sigma_down^2 = sigma_to^2``, and every scheme's fixed point on an exact-denoiser trajectory.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def _alphas_cumprod(num_train_timesteps, beta_start, beta_end):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class _SigmaSpace:
    """Shared by Euler / Euler-ancestral / LMS: sigma_i = sqrt((1 - abar_i) / abar_i), timesteps = linspace(0, T-1, n)[::-1]
    (fractional), sigmas interpolated at them, a trailing 0; the model input is x / sqrt(sigma^2 + 1)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        self.num_train_timesteps = num_train_timesteps
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        self.init_noise_sigma = float(np.concatenate([sig[::-1], [0.0]]).astype(np.float32).max())
        self.num_inference_steps = None

    def set_timesteps(self, n: int) -> np.ndarray:
        self.num_inference_steps = n
        ts = np.linspace(0, self.num_train_timesteps - 1, n, dtype=float)[::-1].copy()
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = ts
        self.derivatives = []
        return ts

    def _index(self, t) -> int:
        return int(np.nonzero(self.timesteps == float(t))[0][0])

    def scale_model_input(self, x, t):
        sigma = self.sigmas[self._index(t)]
        return x / ((sigma ** 2 + 1) ** 0.5)


class EulerOracle(_SigmaSpace):
    def step(self, model_output, t, sample, eta: float = 0.0, noise=None):
        i = self._index(t)
        sigma = self.sigmas[i]
        sigma_hat = sigma                                         # s_churn = 0: gamma = 0
        pred_original_sample = sample - sigma_hat * model_output
        derivative = (sample - pred_original_sample) / sigma_hat
        dt = self.sigmas[i + 1] - sigma_hat
        return sample + derivative * dt


class EulerAncestralOracle(_SigmaSpace):
    def step(self, model_output, t, sample, eta: float = 0.0, noise=None):
        """``noise``: the N(0, 1) draw the dependency makes with ``torch.randn(model_output.shape, generator=...)``."""
        i = self._index(t)
        sigma = self.sigmas[i]
        pred_original_sample = sample - sigma * model_output
        sigma_from, sigma_to = self.sigmas[i], self.sigmas[i + 1]
        sigma_up = (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5
        sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
        derivative = (sample - pred_original_sample) / sigma
        dt = sigma_down - sigma
        prev_sample = sample + derivative * dt
        return prev_sample + noise * sigma_up


def lms_coefficient_exact(sigmas, order: int, t: int, current_order: int) -> float:
    """Integral over [sigma_t, sigma_{t+1}] of the Lagrange basis polynomial prod_{k != j} (tau - s_{t-k}) / (s_{t-j} - s_{t-k}),
    integrated exactly through its polynomial coefficients (the dependency integrates numerically, epsrel = 1e-4)."""
    s = [float(sigmas[t - k]) for k in range(order)]
    poly = np.poly1d([1.0])
    for k in range(order):
        if k == current_order:
            continue
        poly = poly * np.poly1d([1.0, -s[k]]) / (s[current_order] - s[k])
    integ = poly.integ()
    return float(integ(float(sigmas[t + 1])) - integ(float(sigmas[t])))


class LMSOracle(_SigmaSpace):
    def step(self, model_output, t, sample, eta: float = 0.0, noise=None, order: int = 4):
        i = self._index(t)
        sigma = self.sigmas[i]
        pred_original_sample = sample - sigma * model_output
        derivative = (sample - pred_original_sample) / sigma
        self.derivatives.append(derivative)
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(i + 1, order)
        coeffs = [lms_coefficient_exact(self.sigmas, order, i, j) for j in range(order)]
        return sample + sum(c * d for c, d in zip(coeffs, reversed(self.derivatives)))


class DPMSolverPPOracle:
    """``DPMSolverMultistepScheduler(algorithm_type="dpmsolver++", solver_order=2, solver_type="midpoint",
    lower_order_final=True, prediction_type="epsilon")`` -- integer timesteps, init_noise_sigma = 1."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        self.num_train_timesteps = num_train_timesteps
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end)
        self.alpha_t = torch.sqrt(self.alphas_cumprod)
        self.sigma_t = torch.sqrt(1 - self.alphas_cumprod)
        self.lambda_t = torch.log(self.alpha_t) - torch.log(self.sigma_t)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None

    def set_timesteps(self, n: int) -> np.ndarray:
        self.num_inference_steps = n
        self.timesteps = np.linspace(0, self.num_train_timesteps - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        self.model_outputs = [None, None]
        self.lower_order_nums = 0
        return self.timesteps

    def scale_model_input(self, x, t):
        return x

    def _first(self, m0, t, prev, sample):
        lam_t, lam_s = self.lambda_t[prev], self.lambda_t[t]
        alpha_t = self.alpha_t[prev]
        sigma_t, sigma_s = self.sigma_t[prev], self.sigma_t[t]
        h = lam_t - lam_s
        return (sigma_t / sigma_s) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * m0

    def _second(self, outs, tlist, prev, sample):
        t, s0, s1 = prev, tlist[-1], tlist[-2]
        m0, m1 = outs[-1], outs[-2]
        lam_t, lam_s0, lam_s1 = self.lambda_t[t], self.lambda_t[s0], self.lambda_t[s1]
        alpha_t = self.alpha_t[t]
        sigma_t, sigma_s0 = self.sigma_t[t], self.sigma_t[s0]
        h, h_0 = lam_t - lam_s0, lam_s0 - lam_s1
        r0 = h_0 / h
        d0, d1 = m0, (1.0 / r0) * (m0 - m1)
        return (sigma_t / sigma_s0) * sample - (alpha_t * (torch.exp(-h) - 1.0)) * d0 - 0.5 * (alpha_t * (torch.exp(-h) - 1.0)) * d1

    def step(self, model_output, t, sample, eta: float = 0.0, noise=None):
        t = int(t)
        idx = np.nonzero(self.timesteps == t)[0]
        i = len(self.timesteps) - 1 if len(idx) == 0 else int(idx[0])
        prev = 0 if i == len(self.timesteps) - 1 else int(self.timesteps[i + 1])
        lower_final = (i == len(self.timesteps) - 1) and len(self.timesteps) < 15
        x0 = (sample - self.sigma_t[t] * model_output) / self.alpha_t[t]          # dpmsolver++, epsilon prediction
        self.model_outputs = [self.model_outputs[1], x0]
        if self.lower_order_nums < 1 or lower_final:
            out = self._first(x0, t, prev, sample)
        else:
            out = self._second(self.model_outputs, [int(self.timesteps[i - 1]), t], prev, sample)
        if self.lower_order_nums < 2:
            self.lower_order_nums += 1
        return out
