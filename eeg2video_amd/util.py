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


# ---- save_videos_grid (util.py:20-32): the host-side end of the path ------------------------------------
def make_grid(x: torch.Tensor, nrow: int = 8, padding: int = 2, pad_value: float = 0.0) -> torch.Tensor:
    """``torchvision.utils.make_grid`` as ``save_videos_grid`` calls it (util.py:24; torchvision is a dependency of the
    reference, not of this build): ``x [B, C, H, W]`` -> ``[C, ymaps (H + p) + p, xmaps (W + p) + p]``, images in row-major
    order, ``nrow`` images per row, ``pad_value`` between them; a single image is returned as it is."""
    if x.dim() != 4:
        raise ValueError(f"expected [B, C, H, W], got {tuple(x.shape)}")
    if x.shape[1] == 1:
        x = torch.cat((x, x, x), 1)                       # single-channel images become RGB
    nmaps = x.shape[0]
    if nmaps == 1:
        return x[0]
    xmaps = min(nrow, nmaps)
    ymaps = -(-nmaps // xmaps)
    h, w = x.shape[2] + padding, x.shape[3] + padding
    grid = x.new_full((x.shape[1], h * ymaps + padding, w * xmaps + padding), pad_value)
    k = 0
    for yy in range(ymaps):
        for xx in range(xmaps):
            if k >= nmaps:
                break
            grid[:, yy * h + padding: yy * h + h, xx * w + padding: xx * w + w] = x[k]
            k += 1
    return grid


def save_videos_grid(videos: torch.Tensor, path: str, rescale: bool = False, n_rows: int = 4, fps: int = 3):
    """``save_videos_grid(video, f"./{savename}/{i}.gif")`` of ``inference_eeg2video.py:98`` (definition util.py:20-32): videos
    ``[B, 3, T, H, W]`` in [0, 1] -> one grid image per frame -> ``(x * 255).astype(uint8)`` -> an animated GIF at ``fps``.
    The float -> uint8 step runs on the GPU when the videos are still there (``e2v_frames_to_uint8``: 1 byte per sample over
    PCIe instead of 4).  The reference writes through imageio; here the GIF goes through Pillow (what imageio's GIF plugin
    wraps), and a path ending in ``.npy`` stores the uint8 frames ``[T, Hg, Wg, 3]`` as they are.  Returns the uint8 frames."""
    import os

    import numpy as np
    if videos.dim() != 5:
        raise ValueError(f"expected videos [B, C, T, H, W], got {tuple(videos.shape)}")
    frames = videos.permute(2, 0, 1, 3, 4)                                   # "b c t h w -> t b c h w"
    grids = torch.stack([make_grid(x, nrow=n_rows) for x in frames])         # [T, C, Hg, Wg]
    grids = grids.permute(0, 2, 3, 1)                                        # x.transpose(0, 1).transpose(1, 2)
    if rescale and grids.dtype != torch.uint8:
        grids = (grids + 1.0) / 2.0
    if grids.dtype == torch.uint8:          # frames already converted on the device (Engine.frames_to_uint8): laid out only
        out = grids.cpu().numpy()
    elif grids.is_cuda:
        from .engine import Engine
        eng = getattr(save_videos_grid, "engine", None)
        if isinstance(eng, Engine):
            out = eng.frames_to_uint8(grids.contiguous()).cpu().numpy()
        else:
            out = (grids * 255).to(torch.uint8).cpu().numpy()
    else:
        out = (grids.float() * 255).numpy().astype(np.uint8)
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    if path.endswith(".npy"):
        np.save(path, out)
    else:
        from PIL import Image
        imgs = [Image.fromarray(f) for f in out]
        imgs[0].save(path, save_all=True, append_images=imgs[1:], duration=int(round(1000.0 / fps)), loop=0)
    return out
