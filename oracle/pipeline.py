"""Oracle: the denoising loop of ``TuneAVideoPipeline.__call__`` on the CPU.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Follows
``EEG2Video/pipelines/pipeline_tuneeeg2video.py:247-343`` with conditioning handed over
as tensors (the EEG encoder is outside the path).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from .ddim import DDIMOracle
from .unet3d import unet3d_forward
from .vae import vae_decode


def decode_latents(vae_sd, vae_cfg, latents: torch.Tensor) -> torch.Tensor:
    """pipeline_tuneeeg2video.py:175-184 (returns a torch tensor instead of numpy)."""
    b, c, f, h, w = latents.shape
    z = 1 / 0.18215 * latents                                                        # :177
    z = z.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)                             # :178
    video = torch.cat([vae_decode(vae_sd, vae_cfg, z[i:i + 1]) for i in range(b * f)])   # :179 (+ vae slicing)
    video = video.reshape(b, f, video.shape[1], video.shape[2], video.shape[3]).permute(0, 2, 1, 3, 4)
    return (video / 2 + 0.5).clamp(0, 1).float()                                     # :181-183


@torch.no_grad()
def generate(unet_sd, unet_cfg, vae_sd, vae_cfg, latents: torch.Tensor, cond: torch.Tensor,
             uncond: torch.Tensor, num_inference_steps: int = 50, guidance_scale: float = 7.5,
             eta: float = 0.0, trace: Optional[Dict[str, List[torch.Tensor]]] = None,
             decode: bool = True, scheduler=None, noises: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """latents ``[B,4,F,h,w]``, cond ``[B,77,D]``, uncond ``[1 or B,77,D]`` -> videos ``[B,3,F,8h,8w]``.

    ``trace`` (optional) collects per-step ``eps`` (after guidance), the two unguided halves ``eps_u`` / ``eps_c`` and ``latents`` for
    the teacher-forced per-step parity tests; ``trace["taps_step"] = k`` on entry additionally keeps the block outputs of step k's
    UNet forward (``unet3d_forward(..., taps=...)``: emb, down0..3, mid, up0..3) in ``trace["taps"]``.  ``noises`` (optional): the per-step N(0, 1) draws of a stochastic scheduler
    (DDIM with ``eta > 0``, Euler-ancestral), passed in so that a test can hand the device path the same numbers."""
    sched = scheduler if scheduler is not None else DDIMOracle()      # any oracle scheduler (DDIMOracle, PNDMOracle)
    b = latents.shape[0]
    do_cfg = guidance_scale > 1.0                                                    # :281
    if do_cfg:                                                                       # :162-172 (uncond first)
        emb = torch.cat([uncond.expand(b, -1, -1) if uncond.shape[0] == 1 else uncond, cond])
    else:
        emb = cond
    timesteps = sched.set_timesteps(num_inference_steps)                             # :287-288
    x = latents * sched.init_noise_sigma                                             # :244
    for i, t in enumerate(timesteps):                                                # :311
        t = int(t) if float(t).is_integer() else float(t)         # sigma-space schedulers step through fractional timesteps
        x_in = torch.cat([x] * 2) if do_cfg else x                                   # :313
        x_in = sched.scale_model_input(x_in, t)                                      # :314
        taps = {} if trace is not None and trace.get("taps_step") == i else None    # block outputs of this step's forward (tests)
        eps = unet3d_forward(unet_sd, unet_cfg, x_in, t, emb, taps=taps)             # :317
        if taps is not None:
            trace["taps"] = taps
        if do_cfg:                                                                   # :320-322
            eps_u, eps_c = eps.chunk(2)
            if trace is not None:
                trace.setdefault("eps_u", []).append(eps_u.clone())
                trace.setdefault("eps_c", []).append(eps_c.clone())
            eps = eps_u + guidance_scale * (eps_c - eps_u)
        x = sched.step(eps, t, x, eta=eta, noise=noises[i] if noises is not None else None)   # :325
        if trace is not None:
            trace.setdefault("eps", []).append(eps.clone())
            trace.setdefault("latents", []).append(x.clone())
    if not decode:
        return x
    return decode_latents(vae_sd, vae_cfg, x)                                        # :334
