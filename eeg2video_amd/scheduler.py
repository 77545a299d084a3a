"""``DDIMScheduler`` look-alike: host-side schedule, device-side update through the HIP library.

Mirrors what the reference pipeline uses of diffusers' scheduler:
``set_timesteps`` / ``timesteps`` (``pipeline_tuneeeg2video.py:287-288``), ``init_noise_sigma`` (:244),
``scale_model_input`` (:314), ``step(...).prev_sample`` (:325), ``config.steps_offset`` /
``config.clip_sample`` (:59-84), ``order`` (:309), plus ``alphas_cumprod`` / ``final_alpha_cumprod`` /
``config.num_train_timesteps`` which ``tuneavideo/util.py:56-66`` reads.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .unet import FrozenDict


class DDIMSchedulerOutput:
    def __init__(self, prev_sample, pred_original_sample=None):
        self.prev_sample = prev_sample
        self.pred_original_sample = pred_original_sample

    def __getitem__(self, k):
        return getattr(self, k) if isinstance(k, str) else (self.prev_sample,)[k]


class DDIMScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", clip_sample: bool = False, set_alpha_to_one: bool = False,
                 steps_offset: int = 1, engine=None):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        if clip_sample or set_alpha_to_one:
            raise NotImplementedError("clip_sample / set_alpha_to_one are off on this path (pipeline_tuneeeg2video.py:73-84)")
        self._internal_dict = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                         beta_schedule=beta_schedule, clip_sample=clip_sample,
                                         set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.engine = None
        if engine is not None:
            self.bind(engine)

    @property
    def config(self):
        return self._internal_dict

    def bind(self, engine):
        """Attach the HIP engine that executes ``step`` and hand it this host's alpha-bar table."""
        self.engine = engine
        engine.set_ddim_schedule(self.alphas_cumprod.numpy(), self.config.steps_offset)
        return self

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)          # kept on the host: the loop indexes the schedule there

    def scale_model_input(self, sample, timestep=None):
        return sample

    def prev_timestep(self, timestep: int) -> int:
        return int(timestep) - self.config.num_train_timesteps // self.num_inference_steps

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, eta: float = 0.0,
             use_clipped_model_output: bool = False, generator=None, variance_noise=None, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if eta != 0.0:
            raise NotImplementedError("only the deterministic update (eta = 0) is implemented")
        if self.engine is None:
            raise RuntimeError("DDIMScheduler is not bound to an Engine (no CPU path exists)")
        t = int(timestep)
        prev = self.engine.ddim_cfg_step(model_output, None, sample, 1.0, t, self.prev_timestep(t))
        return DDIMSchedulerOutput(prev) if return_dict else (prev,)


class PNDMScheduler:
    """``PNDMScheduler`` look-alike (diffusers 0.11.1; stock SD-v1-4 ``scheduler_config.json``: ``skip_prk_steps=True``,
    ``steps_offset=1``, ``set_alpha_to_one=False``, scaled-linear betas), i.e. the PLMS linear multistep scheme -- the first
    of the non-DDIM schedulers ``TuneAVideoPipeline.__init__`` accepts (``pipeline_tuneeeg2video.py:48-55``).  Host side
    keeps the schedule, the step counter and the list of the last four model outputs (device tensors); the arithmetic runs
    in the HIP library (``e2v_lincomb``).  Dependency-owned algorithm: parity unpinned by the reference, checked against
    ``oracle/pndm.py`` and closed forms."""
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", skip_prk_steps: bool = True, set_alpha_to_one: bool = False,
                 steps_offset: int = 1, engine=None):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        if not skip_prk_steps or set_alpha_to_one:
            raise NotImplementedError("only the Stable-Diffusion configuration (skip_prk_steps, no alpha-to-one) is implemented")
        self._internal_dict = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                         beta_schedule=beta_schedule, skip_prk_steps=skip_prk_steps,
                                         set_alpha_to_one=set_alpha_to_one, steps_offset=steps_offset)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.pndm_order = 4
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.counter, self.cur_sample, self.ets = 0, None, []
        self.engine = engine

    @property
    def config(self):
        return self._internal_dict

    def bind(self, engine):
        self.engine = engine
        return self

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        base = (np.arange(0, num_inference_steps) * ratio).round().astype(np.int64) + self.config.steps_offset
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()        # second-to-last timestep twice
        self.timesteps = torch.from_numpy(plms.astype(np.int64))
        self.counter, self.cur_sample, self.ets = 0, None, []

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _prev_sample(self, sample, t: int, prev: int, model_output):
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        b_t, b_p = 1 - a_t, 1 - a_p
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * b_p ** 0.5 + (a_t * b_t * a_p) ** 0.5
        return self.engine.lincomb([(float(sample_coeff), sample), (float(-(a_p - a_t) / denom), model_output)])

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self.engine is None:
            raise RuntimeError("PNDMScheduler is not bound to an Engine (no CPU path exists)")
        t = int(timestep)
        ratio = self.config.num_train_timesteps // self.num_inference_steps
        prev = t - ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev, t = t, t + ratio
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(e) == 1 and self.counter == 1:
            model_output = self.engine.lincomb([(0.5, model_output), (0.5, e[-1])])
            sample, self.cur_sample = self.cur_sample, None
        elif len(e) == 2:
            model_output = self.engine.lincomb([(1.5, e[-1]), (-0.5, e[-2])])
        elif len(e) == 3:
            model_output = self.engine.lincomb([(23 / 12, e[-1]), (-16 / 12, e[-2]), (5 / 12, e[-3])])
        else:
            model_output = self.engine.lincomb([(55 / 24, e[-1]), (-59 / 24, e[-2]), (37 / 24, e[-3]), (-9 / 24, e[-4])])
        prev_sample = self._prev_sample(sample, t, prev, model_output)
        self.counter += 1
        return DDIMSchedulerOutput(prev_sample) if return_dict else (prev_sample,)
