"""Oracle: ``PNDMScheduler`` in the Stable-Diffusion configuration (PLMS), torch CPU fp32.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  The scheduler belongs to the absent dependency ``diffusers==0.11.1``
(one of the types ``TuneAVideoPipeline.__init__`` accepts, ``EEG2Video/pipelines/pipeline_tuneeeg2video.py:48-55``);
restated from its published algorithm (``set_timesteps`` with ``skip_prk_steps``, ``step_plms``, ``_get_prev_sample``,
formula (9) of the PNDM paper) -- PARITY UNPINNED by the reference, anchored on closed forms: the timestep lists
(n = 4 -> 751, 501, 501, 251, 1), coefficient sums of the multistep combinations, and agreement of the first step with the
DDIM update it degenerates to.
"""
from __future__ import annotations

import numpy as np
import torch


class PNDMOracle:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.num_train_timesteps, self.steps_offset = num_train_timesteps, steps_offset
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]          # set_alpha_to_one = False
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None

    def set_timesteps(self, n: int) -> np.ndarray:
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        base = (np.arange(0, n) * ratio).round().astype(np.int64) + self.steps_offset
        self.timesteps = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        self.counter, self.cur_sample, self.ets = 0, None, []
        return self.timesteps

    def scale_model_input(self, x, t):
        return x

    def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
        alpha_prod_t = self.alphas_cumprod[timestep]
        alpha_prod_t_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        beta_prod_t = 1 - alpha_prod_t
        beta_prod_t_prev = 1 - alpha_prod_t_prev
        sample_coeff = (alpha_prod_t_prev / alpha_prod_t) ** 0.5
        model_output_denom_coeff = alpha_prod_t * beta_prod_t_prev ** 0.5 + (alpha_prod_t * beta_prod_t * alpha_prod_t_prev) ** 0.5
        return sample_coeff * sample - (alpha_prod_t_prev - alpha_prod_t) * model_output / model_output_denom_coeff

    def step(self, model_output: torch.Tensor, timestep: int, sample: torch.Tensor, eta: float = 0.0, noise=None) -> torch.Tensor:
        timestep = int(timestep)
        ratio = self.num_train_timesteps // self.num_inference_steps
        prev_timestep = timestep - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_timestep = timestep
            timestep = timestep + ratio
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        prev_sample = self._get_prev_sample(sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return prev_sample
