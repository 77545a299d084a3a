"""Oracle: the DDIM scheduler as the pipeline uses it.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  ``DDIMScheduler`` belongs to the absent
dependency ``diffusers==0.11.1``; restated from its published algorithm
with the SD-v1-4 scheduler config -- PARITY UNPINNED by the reference, anchored on:

* the reference's call sites ``EEG2Video/pipelines/pipeline_tuneeeg2video.py:287,314,325``
  and its config patches ``:59-84`` (``steps_offset = 1``, ``clip_sample = False``);
* the in-repo restatement of the same update ``EEG2Video_New/Generation/tuneavideo/util.py:56-66``
  (``alphas_cumprod``, ``final_alpha_cumprod``, ``num_train_timesteps // num_inference_steps``);
* closed forms: n=50 -> 981, 961, ..., 21, 1; n=4 -> 751, 501, 251, 1.
"""
from __future__ import annotations

import numpy as np
import torch


class DDIMOracle:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]          # set_alpha_to_one = False
        self.init_noise_sigma = 1.0
        self.timesteps = None
        self.num_inference_steps = None

    def set_timesteps(self, n: int) -> np.ndarray:
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.steps_offset
        self.timesteps = ts
        return ts

    def scale_model_input(self, x, t):
        return x

    def step(self, eps: torch.Tensor, t: int, x: torch.Tensor, eta: float = 0.0, noise=None) -> torch.Tensor:
        """``eta > 0`` (formulas (12), (16) of the DDIM paper): ``noise`` is the N(0, 1) draw the dependency makes with
        ``torch.randn(model_output.shape, generator=...)`` (call site pipeline_tuneeeg2video.py:306,325)."""
        t = int(t)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        beta_t = 1 - a_t
        x0 = (x - beta_t ** 0.5 * eps) / a_t ** 0.5
        var = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)
        std = eta * var ** 0.5
        direction = (1 - a_p - std ** 2) ** 0.5 * eps
        prev = a_p ** 0.5 * x0 + direction
        if eta > 0:
            prev = prev + std * noise
        return prev


# ---- DDIM inversion: EEG2Video_New/Generation/tuneavideo/util.py:56-101 (reference-owned; PINNED by
# tests/golden/reference_t1_inversion.npz, generated from the reference's own next_step) -----------------
def next_step(model_output: torch.Tensor, timestep: int, sample: torch.Tensor, sched: DDIMOracle) -> torch.Tensor:
    """util.py:56-66, line for line."""
    timestep, next_timestep = min(int(timestep) - sched.num_train_timesteps // sched.num_inference_steps, 999), int(timestep)
    alpha_prod_t = sched.alphas_cumprod[timestep] if timestep >= 0 else sched.final_alpha_cumprod
    alpha_prod_t_next = sched.alphas_cumprod[next_timestep]
    beta_prod_t = 1 - alpha_prod_t
    next_original_sample = (sample - beta_prod_t ** 0.5 * model_output) / alpha_prod_t ** 0.5
    next_sample_direction = (1 - alpha_prod_t_next) ** 0.5 * model_output
    return alpha_prod_t_next ** 0.5 * next_original_sample + next_sample_direction


def ddim_loop(unet_fn, sched: DDIMOracle, latent: torch.Tensor, num_inv_steps: int, cond: torch.Tensor):
    """util.py:74-93 with the cond embeddings passed in (the reference reads them from ``cond_embeddings.pt``, :80-82).
    ``unet_fn(latent, t, cond) -> eps``."""
    cond = cond.repeat(latent.shape[0], 1, 1) if cond.shape[0] == 1 and latent.shape[0] > 1 else cond
    all_latent = [latent]
    latent = latent.clone()
    for i in range(num_inv_steps):
        t = sched.timesteps[len(sched.timesteps) - i - 1]
        noise_pred = unet_fn(latent, int(t), cond)
        latent = next_step(noise_pred, int(t), latent, sched)
        all_latent.append(latent)
    return all_latent
