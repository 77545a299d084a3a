"""``DDIMScheduler`` look-alike: host-side schedule, device-side update through the HIP library.

Mirrors what the reference pipeline uses of diffusers' scheduler:
``set_timesteps`` / ``timesteps`` (``pipeline_tuneeeg2video.py:287-288``), ``init_noise_sigma`` (:244),
``scale_model_input`` (:314), ``step(...).prev_sample`` (:325), ``config.steps_offset`` /
``config.clip_sample`` (:59-84), ``order`` (:309), plus ``alphas_cumprod`` / ``final_alpha_cumprod`` /
``config.num_train_timesteps`` which ``tuneavideo/util.py:56-66`` reads.
"""
from __future__ import annotations

import math
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
        if self.engine is None:
            raise RuntimeError("DDIMScheduler is not bound to an Engine (no CPU path exists)")
        if use_clipped_model_output:
            raise NotImplementedError("use_clipped_model_output is off on this path (clip_sample = False)")
        if eta != 0.0:
            prev = _ddim_step_eta(self, model_output, timestep, sample, float(eta), generator, variance_noise)
            return DDIMSchedulerOutput(prev) if return_dict else (prev,)
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


def _randn_like(model_output: torch.Tensor, generator=None) -> torch.Tensor:
    """The N(0, 1) draw the dependency's schedulers make: ``torch.randn(model_output.shape, generator=generator, ...)`` on the
    generator's device (a CPU generator gives the same numbers as in the reference run), then moved next to the sample."""
    gdev = generator.device if generator is not None else model_output.device
    return torch.randn(model_output.shape, generator=generator, device=gdev, dtype=torch.float32).to(model_output.device)


def _ddim_step_eta(self, model_output, timestep, sample, eta, generator=None, variance_noise=None):
    """Stochastic DDIM (``eta > 0``; call site ``pipeline_tuneeeg2video.py:306,325``): one three-term ``e2v_lincomb``
    x_prev = (sqrt(a_p) / sqrt(a_t)) x + (sqrt(1 - a_p - s^2) - sqrt(a_p) sqrt(1 - a_t) / sqrt(a_t)) eps + s z."""
    t = int(timestep)
    prev = self.prev_timestep(t)
    a_t = float(self.alphas_cumprod[t])
    a_p = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
    var = (1 - a_p) / (1 - a_t) * (1 - a_t / a_p)
    std = eta * var ** 0.5
    z = variance_noise if variance_noise is not None else _randn_like(model_output, generator)
    cx = a_p ** 0.5 / a_t ** 0.5
    ce = (1 - a_p - std ** 2) ** 0.5 - a_p ** 0.5 * (1 - a_t) ** 0.5 / a_t ** 0.5
    return self.engine.lincomb([(cx, sample), (ce, model_output), (std, z)])


class _SigmaSpaceScheduler:
    """Common part of the k-diffusion style schedulers (``EulerDiscreteScheduler``, ``EulerAncestralDiscreteScheduler``,
    ``LMSDiscreteScheduler`` of diffusers 0.11.1): sigma_i = sqrt((1 - abar_i) / abar_i); ``set_timesteps`` takes
    ``linspace(0, T-1, n)[::-1]`` (fractional) and interpolates the sigmas there, appending 0; the model sees
    ``x / sqrt(sigma^2 + 1)`` and the latents start at ``init_noise_sigma = max sigma``.  Host side: tables and bookkeeping;
    device side: ``e2v_lincomb``.  Dependency-owned: parity unpinned, checked against ``oracle/schedulers.py``."""
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", prediction_type: str = "epsilon", engine=None, **ignored):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError(f"{beta_schedule} does is not implemented for {self.__class__}")
        if prediction_type != "epsilon":
            raise NotImplementedError("only epsilon prediction is used on this path")
        self._internal_dict = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                         beta_schedule=beta_schedule, prediction_type=prediction_type)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.concatenate([sig[::-1], [0.0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sig)
        self.init_noise_sigma = self.sigmas.max()
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=float)[::-1].copy())
        self.derivatives = []
        self.is_scale_input_called = False
        self.engine = engine

    @property
    def config(self):
        return self._internal_dict

    def bind(self, engine):
        self.engine = engine
        return self

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(0, self.config.num_train_timesteps - 1, num_inference_steps, dtype=float)[::-1].copy()
        sig = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)             # float64, host
        self.derivatives = []

    def _index(self, timestep) -> int:
        t = float(timestep)
        hit = (self.timesteps == t).nonzero()
        if hit.numel() == 0:
            raise ValueError(f"timestep {t} is not one of the scheduler's timesteps")
        return int(hit[0].item())

    def scale_model_input(self, sample, timestep):
        if self.engine is None:
            raise RuntimeError(f"{self.__class__.__name__} is not bound to an Engine (no CPU path exists)")
        sigma = float(self.sigmas[self._index(timestep)])
        self.is_scale_input_called = True
        return self.engine.lincomb([(1.0 / (sigma ** 2 + 1) ** 0.5, sample)])

    def _check_step(self):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self.engine is None:
            raise RuntimeError(f"{self.__class__.__name__} is not bound to an Engine (no CPU path exists)")


class EulerDiscreteScheduler(_SigmaSpaceScheduler):
    """Algorithm 2 (Euler steps) of Karras et al. 2022 with s_churn = 0: x' = x + (sigma_{i+1} - sigma_i) eps."""

    def step(self, model_output, timestep, sample, s_churn: float = 0.0, s_tmin: float = 0.0, s_tmax: float = float("inf"),
             s_noise: float = 1.0, generator=None, return_dict: bool = True):
        self._check_step()
        if s_churn != 0.0:
            raise NotImplementedError("s_churn > 0 (stochastic Euler) is not used by the pipeline")
        i = self._index(timestep)
        sigma, sigma_next = float(self.sigmas[i]), float(self.sigmas[i + 1])
        prev = self.engine.lincomb([(1.0, sample), (sigma_next - sigma, model_output)])      # derivative = eps (epsilon prediction)
        return DDIMSchedulerOutput(prev) if return_dict else (prev,)


class EulerAncestralDiscreteScheduler(_SigmaSpaceScheduler):
    """Ancestral sampling with Euler steps (k-diffusion ``sample_euler_ancestral``): a deterministic step to sigma_down plus
    fresh noise of scale sigma_up."""

    def step(self, model_output, timestep, sample, generator=None, return_dict: bool = True, noise=None):
        self._check_step()
        i = self._index(timestep)
        s_from, s_to = float(self.sigmas[i]), float(self.sigmas[i + 1])
        s_up = (s_to ** 2 * (s_from ** 2 - s_to ** 2) / s_from ** 2) ** 0.5
        s_down = (s_to ** 2 - s_up ** 2) ** 0.5
        z = noise if noise is not None else _randn_like(model_output, generator)
        prev = self.engine.lincomb([(1.0, sample), (s_down - s_from, model_output), (s_up, z)])
        return DDIMSchedulerOutput(prev) if return_dict else (prev,)


class LMSDiscreteScheduler(_SigmaSpaceScheduler):
    """Linear multistep (Adams-Bashforth in sigma, k-diffusion ``sample_lms``), order 4: the coefficients are integrals of the
    Lagrange basis over [sigma_i, sigma_{i+1}], evaluated numerically with the dependency's own call
    (``scipy.integrate.quad(..., epsrel=1e-4)``)."""

    def get_lms_coefficient(self, order: int, t: int, current_order: int) -> float:
        from scipy import integrate

        def lms_derivative(tau):
            prod = 1.0
            for k in range(order):
                if current_order == k:
                    continue
                prod *= (tau - self.sigmas[t - k]) / (self.sigmas[t - current_order] - self.sigmas[t - k])
            return prod

        return float(integrate.quad(lms_derivative, self.sigmas[t], self.sigmas[t + 1], epsrel=1e-4)[0])

    def step(self, model_output, timestep, sample, order: int = 4, return_dict: bool = True):
        self._check_step()
        i = self._index(timestep)
        self.derivatives.append(model_output)                    # derivative = (x - (x - sigma eps)) / sigma = eps
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(i + 1, order)
        coeffs = [self.get_lms_coefficient(order, i, j) for j in range(order)]
        terms = [(1.0, sample)] + [(c, d) for c, d in zip(coeffs, reversed(self.derivatives))]
        prev = self.engine.lincomb(terms)
        return DDIMSchedulerOutput(prev) if return_dict else (prev,)


class DPMSolverMultistepScheduler:
    """DPM-Solver++ (2M) as Stable Diffusion configures it in diffusers 0.11.1: ``algorithm_type="dpmsolver++"``,
    ``solver_order=2``, ``solver_type="midpoint"``, ``lower_order_final=True``, epsilon prediction, no thresholding.  The data
    prediction x0 = (x - sigma_t eps) / alpha_t is kept for one step; an update is one ``e2v_lincomb`` of (x, x0, x0_prev)."""
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012,
                 beta_schedule: str = "scaled_linear", solver_order: int = 2, prediction_type: str = "epsilon",
                 thresholding: bool = False, algorithm_type: str = "dpmsolver++", solver_type: str = "midpoint",
                 lower_order_final: bool = True, engine=None, **ignored):
        if beta_schedule != "scaled_linear" or prediction_type != "epsilon" or thresholding:
            raise NotImplementedError("only the Stable-Diffusion configuration (scaled_linear, epsilon, no thresholding) is implemented")
        if algorithm_type != "dpmsolver++" or solver_type != "midpoint" or solver_order not in (1, 2):
            raise NotImplementedError("only dpmsolver++ with the midpoint rule, order 1 or 2, is implemented")
        self._internal_dict = FrozenDict(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                         beta_schedule=beta_schedule, solver_order=solver_order, prediction_type=prediction_type,
                                         thresholding=thresholding, algorithm_type=algorithm_type, solver_type=solver_type,
                                         lower_order_final=lower_order_final)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.alpha_t = torch.sqrt(self.alphas_cumprod)
        self.sigma_t = torch.sqrt(1 - self.alphas_cumprod)
        self.lambda_t = torch.log(self.alpha_t) - torch.log(self.sigma_t)
        self.init_noise_sigma = 1.0
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=np.float32)[::-1].copy())
        self.model_outputs = [None] * solver_order
        self.lower_order_nums = 0
        self.engine = engine

    @property
    def config(self):
        return self._internal_dict

    def bind(self, engine):
        self.engine = engine
        return self

    def set_timesteps(self, num_inference_steps: int, device=None):
        self.num_inference_steps = num_inference_steps
        ts = np.linspace(0, self.config.num_train_timesteps - 1, num_inference_steps + 1).round()[::-1][:-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)
        self.model_outputs = [None] * self.config.solver_order
        self.lower_order_nums = 0

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, return_dict: bool = True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        if self.engine is None:
            raise RuntimeError("DPMSolverMultistepScheduler is not bound to an Engine (no CPU path exists)")
        t = int(timestep)
        hit = (self.timesteps == t).nonzero()
        i = len(self.timesteps) - 1 if hit.numel() == 0 else int(hit[0].item())
        prev = 0 if i == len(self.timesteps) - 1 else int(self.timesteps[i + 1])
        lower_final = (i == len(self.timesteps) - 1) and self.config.lower_order_final and len(self.timesteps) < 15
        a_s, s_s = float(self.alpha_t[t]), float(self.sigma_t[t])
        x0 = self.engine.lincomb([(1.0 / a_s, sample), (-s_s / a_s, model_output)])          # convert_model_output
        self.model_outputs = self.model_outputs[1:] + [x0]
        lam_t, lam_s = float(self.lambda_t[prev]), float(self.lambda_t[t])
        a_t, s_t = float(self.alpha_t[prev]), float(self.sigma_t[prev])
        h = lam_t - lam_s
        cm = a_t * (math.exp(-h) - 1.0)
        if self.config.solver_order == 1 or self.lower_order_nums < 1 or lower_final:
            out = self.engine.lincomb([(s_t / s_s, sample), (-cm, x0)])
        else:
            s1 = int(self.timesteps[i - 1])
            h0 = lam_s - float(self.lambda_t[s1])
            r0 = h0 / h
            m1 = self.model_outputs[-2]
            # x_t = (s_t / s_s) x - cm D0 - 0.5 cm D1,  D0 = m0,  D1 = (m0 - m1) / r0
            out = self.engine.lincomb([(s_t / s_s, sample), (-cm * (1.0 + 0.5 / r0), x0), (0.5 * cm / r0, m1)])
        if self.lower_order_nums < self.config.solver_order:
            self.lower_order_nums += 1
        return DDIMSchedulerOutput(out) if return_dict else (out,)


#: ``_class_name`` of a diffusers ``scheduler_config.json`` -> mirror class (the six types the pipeline's constructor accepts)
SCHEDULERS = {
    "DDIMScheduler": DDIMScheduler, "PNDMScheduler": PNDMScheduler, "LMSDiscreteScheduler": LMSDiscreteScheduler,
    "EulerDiscreteScheduler": EulerDiscreteScheduler, "EulerAncestralDiscreteScheduler": EulerAncestralDiscreteScheduler,
    "DPMSolverMultistepScheduler": DPMSolverMultistepScheduler,
}


def scheduler_from_config(config: dict, engine=None):
    """Build the mirror a ``scheduler/scheduler_config.json`` names.  Keys that only describe the file (``_class_name``,
    ``_diffusers_version``) are dropped; a key that CHANGES THE ARITHMETIC and is set to something this build does not implement
    raises ``NotImplementedError`` whether or not the mirror's constructor knows the key (a ``v_prediction`` DDIM config must
    not load silently as epsilon prediction); any other key the mirror does not model is dropped."""
    import inspect
    name = config.get("_class_name", "DDIMScheduler")
    if name not in SCHEDULERS:
        raise ValueError(f"scheduler class {name!r} is not one of {sorted(SCHEDULERS)}")
    cls = SCHEDULERS[name]
    #: key -> the only value(s) implemented (diffusers 0.11.1 defaults of the Stable-Diffusion configs)
    only = {"prediction_type": ("epsilon",), "trained_betas": (None,), "thresholding": (False,), "set_alpha_to_one": (False,),
            "beta_schedule": ("scaled_linear",), "variance_type": (None, "fixed_small"),
            "rescale_betas_zero_snr": (False,), "use_karras_sigmas": (False,),
            # the spacing each 0.11.1 class implements (and newer diffusers write into its config as the default): DDIM / PNDM step
            # through multiples of T // n ("leading"), the sigma-space schedulers and DPM-Solver++ through linspace(0, T - 1, .)
            "timestep_spacing": ("leading",) if name in ("DDIMScheduler", "PNDMScheduler") else ("linspace",)}
    for k, ok in only.items():
        if k in config and config[k] not in ok:
            raise NotImplementedError(f"{name}: {k}={config[k]!r} is not implemented on this path (supported: {ok})")
    params = inspect.signature(cls.__init__).parameters
    kw = {k: v for k, v in config.items() if k in params and not k.startswith("_")}
    if kw.get("clip_sample"):          # the pipeline's constructor overrides it (pipeline_tuneeeg2video.py:73-84: "clip_sample ... set to False")
        kw["clip_sample"] = False
    return cls(engine=engine, **kw)


def scheduler_from_pretrained(path: str, subfolder: Optional[str] = "scheduler", engine=None):
    import json
    import os
    p = os.path.join(path, subfolder) if subfolder else path
    f = os.path.join(p, "scheduler_config.json")
    if not os.path.isfile(f):
        raise RuntimeError(f"{f} does not exist")
    with open(f) as fh:
        return scheduler_from_config(json.load(fh), engine=engine)
