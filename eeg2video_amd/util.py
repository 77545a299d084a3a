"""Mirror of the DDIM-inversion helpers of ``EEG2Video_New/Generation/tuneavideo/util.py`` (lines 56-101), the
reference's way of producing the start latents of a validation sample (``train_finetune_videodiffusion.py:326-328``,
fed back through ``pipeline(..., latents=ddim_inv_latent)``).  Same names, argument order and return types; the
arithmetic runs in the HIP library (``e2v_ddim_next_step`` / ``e2v_ddim_invert``), there is no CPU fallback.

Difference to the reference, on purpose: ``ddim_loop`` there ignores ``prompt`` and reads ``cond_embeddings.pt`` from the
working directory (util.py:80-82); here ``prompt`` may BE the cond embeddings ``[1 or B, 77, 768]`` (what the
commented-out lines :84-86 did) and the file is only read when ``prompt`` is a string, as the reference does.
"""
from typing import List, Union

import torch


def _engine_of(unet):
    eng = getattr(unet, "engine", None)
    if eng is None:
        raise TypeError("unet must be an eeg2video_amd UNet3DConditionModel (it carries the HIP engine)")
    return eng


def next_step(model_output: torch.Tensor, timestep: int, sample: torch.Tensor, ddim_scheduler) -> torch.Tensor:
    """util.py:56-66.  ``ddim_scheduler`` must be bound to an engine (``DDIMScheduler.bind``) and have
    ``set_timesteps`` called, exactly as the reference requires ``num_inference_steps`` to be set."""
    eng = getattr(ddim_scheduler, "engine", None)
    if eng is None:
        raise RuntimeError("the scheduler is not bound to a HIP engine (DDIMScheduler.bind(engine))")
    if ddim_scheduler.num_inference_steps is None:
        raise ValueError("call ddim_scheduler.set_timesteps(n) first")
    return eng.ddim_next_step(model_output, int(timestep), sample, int(ddim_scheduler.num_inference_steps))


def get_noise_pred_single(latents, t, context, unet):
    """util.py:68-70."""
    return unet(latents, t, encoder_hidden_states=context)["sample"]


@torch.no_grad()
def ddim_loop(unet, ddim_scheduler, latent: torch.Tensor, num_inv_steps: int, prompt: Union[str, torch.Tensor]) -> List[torch.Tensor]:
    """util.py:74-93: returns ``[latent_0, ..., latent_n]``.  The scheduler's ``timesteps`` must have been set with
    ``set_timesteps(num_inv_steps)`` (train_finetune_videodiffusion.py:202), which the fused device loop assumes."""
    eng = _engine_of(unet)
    if isinstance(prompt, torch.Tensor):
        cond_embeddings = prompt
    else:
        cond_embeddings = torch.load("cond_embeddings.pt", map_location="cpu")          # util.py:80
    cond_embeddings = cond_embeddings.to(device=eng.device, dtype=torch.float32)
    if cond_embeddings.dim() == 2:
        cond_embeddings = cond_embeddings[None]
    if cond_embeddings.shape[0] == 1 and latent.shape[0] > 1:
        cond_embeddings = cond_embeddings.repeat(latent.shape[0], 1, 1)                  # util.py:82
    if ddim_scheduler.num_inference_steps != num_inv_steps:
        raise ValueError(f"ddim_scheduler.set_timesteps({ddim_scheduler.num_inference_steps}) does not match "
                         f"num_inv_steps={num_inv_steps}")
    return eng.ddim_invert(latent, cond_embeddings.contiguous(), num_inv_steps, return_all=True)


@torch.no_grad()
def ddim_inversion(unet, ddim_scheduler, video_latent: torch.Tensor, num_inv_steps: int, prompt: Union[str, torch.Tensor] = ""):
    """util.py:96-99."""
    return ddim_loop(unet, ddim_scheduler, video_latent, num_inv_steps, prompt)
