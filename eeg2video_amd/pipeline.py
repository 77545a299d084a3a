"""``TuneAVideoPipeline`` with the reference's call signature, executed by the HIP library.

Mirror of ``EEG2Video/pipelines/pipeline_tuneeeg2video.py:40-343`` (and the ``EEG2Video_New`` variant
``Generation/pipelines/pipeline_tuneeeg2video.py:155-170`` for ``model=None``): same ``__call__`` kwargs,
same checks and error types, same output object.  What differs, on purpose:

* conditioning stays fp32 (the reference casts to fp16, ``:150``);
* the unconditional embedding is taken from ``negative_prompt`` (a ``[1 or B,77,768]`` tensor) or
  ``self.negative_embeddings`` instead of the hard-coded ``negative.npy`` path of ``:170``, and is
  broadcast to the batch (the reference is only consistent for B = 1);
* when no per-step callback is requested and UNet and VAE share one engine the whole loop runs
  inside ``e2v_generate`` with no host round trip per step; otherwise the loop below steps through
  ``e2v_unet_forward`` / ``e2v_ddim_cfg_step`` exactly as the reference's Python loop does.
"""
from __future__ import annotations

import inspect
from contextlib import contextmanager
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from .scheduler import (DDIMScheduler, DPMSolverMultistepScheduler, EulerAncestralDiscreteScheduler,  # noqa: F401
                        EulerDiscreteScheduler, LMSDiscreteScheduler, PNDMScheduler, scheduler_from_pretrained)
from .unet import UNet3DConditionModel
from .vae import AutoencoderKL


class TuneAVideoPipelineOutput:
    """``BaseOutput`` look-alike (pipeline_tuneeeg2video.py:35-37)."""

    def __init__(self, videos):
        self.videos = videos

    def __getitem__(self, k):
        return self.videos if k in ("videos", 0) else (_ for _ in ()).throw(KeyError(k))


class TuneAVideoPipeline:
    def __init__(self, vae: AutoencoderKL, tokenizer, unet: UNet3DConditionModel, scheduler: DDIMScheduler):
        if getattr(scheduler.config, "steps_offset", 1) != 1:                                    # :59-71
            scheduler._internal_dict["steps_offset"] = 1
        if getattr(scheduler.config, "clip_sample", False):                                      # :73-84
            scheduler._internal_dict["clip_sample"] = False
        self.vae, self.tokenizer, self.unet, self.scheduler = vae, tokenizer, unet, scheduler
        self.vae_scale_factor = 2 ** (len(self.vae.config.block_out_channels) - 1)               # :113
        self.negative_embeddings: Optional[torch.Tensor] = None
        if scheduler.engine is None:
            scheduler.bind(unet.engine)
        self._progress = True

    @classmethod
    def from_pretrained(cls, pretrained_model_path: str, unet: Optional[UNet3DConditionModel] = None, vae: Optional[AutoencoderKL] = None,
                        scheduler=None, tokenizer=None, torch_dtype=None, device: int = 0, **kwargs):
        """``TuneAVideoPipeline.from_pretrained(pretrained_model_path, unet=unet, torch_dtype=torch.float16)``
        (``inference_eeg2video.py:70``) for a LOCAL Stable-Diffusion directory: ``model_index.json`` names the components,
        ``vae/`` (config.json + weights) and ``scheduler/scheduler_config.json`` (DDIM in the tuned checkpoints, PNDM in the
        stock SD-v1-4 one; any of the six types the constructor accepts) are loaded here, components passed in are used as
        they are.  The VAE is created on the UNet's engine so that the fused device loop applies.  Checkpoints of any float type are
        widened to fp32 at load; ``torch_dtype`` selects the ARITHMETIC of the shared engine as it does for the reference pipeline:
        ``torch.float16`` (the reference script) -> the fp16 mode, ``torch.bfloat16`` -> the bf16 mode, ``torch.float32`` -> fp32,
        ``None`` -> whatever the UNet passed in was set to (``pipe.unet.engine.set_compute_dtype`` changes it afterwards)."""
        import json
        import os
        index_file = os.path.join(pretrained_model_path, "model_index.json")
        index = {}
        if os.path.isfile(index_file):
            with open(index_file) as f:
                index = json.load(f)
        elif not os.path.isdir(pretrained_model_path):
            raise RuntimeError(f"{pretrained_model_path} does not exist")
        if unet is None:
            vcfg = vae.vcfg if vae is not None else AutoencoderKL.config_from_dir(os.path.join(pretrained_model_path, "vae"))
            unet = UNet3DConditionModel.from_pretrained(pretrained_model_path, subfolder="unet", torch_dtype=torch_dtype, device=device,
                                                        vae_config=vcfg)
        if vae is None:
            vae = AutoencoderKL.from_pretrained(pretrained_model_path, subfolder="vae", torch_dtype=torch_dtype, engine=unet.engine)
        if scheduler is None:
            scheduler = scheduler_from_pretrained(pretrained_model_path, "scheduler", engine=unet.engine)
            named = index.get("scheduler")
            if isinstance(named, (list, tuple)) and len(named) == 2 and named[1] != type(scheduler).__name__:
                raise RuntimeError(f"model_index.json names the scheduler {named[1]}, scheduler_config.json builds {type(scheduler).__name__}")
        if tokenizer is None and os.path.isdir(os.path.join(pretrained_model_path, "tokenizer")):
            try:                                   # unused on the EEG path (the conditioning is the Semantic Predictor's output)
                from transformers import CLIPTokenizer
                tokenizer = CLIPTokenizer.from_pretrained(os.path.join(pretrained_model_path, "tokenizer"))
            except Exception:
                tokenizer = None
        if torch_dtype is not None:
            unet.to(torch_dtype)
        return cls(vae=vae, tokenizer=tokenizer, unet=unet, scheduler=scheduler)

    # -- small API of DiffusionPipeline the callers use ------------------------------------------------
    @property
    def device(self):
        return self.unet.device

    _execution_device = device

    def to(self, *a, **k):                  # a device is a no-op; a floating dtype selects the arithmetic (see UNet3DConditionModel.to)
        self.unet.to(*a, **k)
        return self

    def enable_vae_slicing(self):                                                                # :115-116
        self.vae.enable_slicing()

    def disable_vae_slicing(self):                                                               # :118-119
        self.vae.disable_slicing()

    def enable_xformers_memory_efficient_attention(self):       # attention is always fused on this path
        return None

    def enable_sequential_cpu_offload(self, gpu_id=0):
        """:121-131 (accelerate's ``cpu_offload`` of the UNet and the VAE).  The weights of this pipeline live in the library's
        context on the GPU (3.6 GB fp32 UNet + 0.3 GB VAE of 288 GB) and there are no torch modules to offload: accepted so
        that a caller which sets it keeps working, checked for the one thing that can be wrong, and without effect."""
        if int(gpu_id) != self.unet.engine.device.index:
            raise ValueError(f"the pipeline's context lives on cuda:{self.unet.engine.device.index}, not on cuda:{gpu_id}")
        return None

    def enable_attention_slicing(self, slice_size="auto"):      # DiffusionPipeline API: forwards to the UNet's (no-op) knob
        self.unet.set_attention_slice(slice_size)

    def disable_attention_slicing(self):
        self.unet.set_attention_slice(None)

    def set_progress_bar_config(self, disable: bool = False, **kw):
        self._progress = not disable

    @contextmanager
    def progress_bar(self, total=None):
        class _Bar:
            def update(self, n=1):
                pass
        bar = _Bar()
        if self._progress:
            try:
                from tqdm.auto import tqdm
                bar = tqdm(total=total)
            except Exception:
                pass
        try:
            yield bar
        finally:
            if hasattr(bar, "close"):
                bar.close()

    # -- reference methods ---------------------------------------------------------------------------
    def _encode_eeg(self, model, eeg, device, num_videos_per_eeg, do_classifier_guidance, negative_eeg):
        """:147-173.  ``model=None``: ``eeg`` already holds the embeddings (New variant :155-170)."""
        emb = model(eeg.to(device)) if model is not None else eeg.to(device)
        emb = torch.reshape(emb, [emb.shape[0], 77, -1]).float()                                 # :150
        bs, seq_len, _ = emb.shape
        emb = emb.repeat(1, num_videos_per_eeg, 1).view(bs * num_videos_per_eeg, seq_len, -1)     # :159-160
        if do_classifier_guidance:
            neg = negative_eeg if negative_eeg is not None else self.negative_embeddings
            if isinstance(neg, np.ndarray):
                neg = torch.from_numpy(neg)
            if not isinstance(neg, torch.Tensor):
                raise ValueError("classifier-free guidance needs the unconditional embedding: pass it as "
                                 "`negative_prompt` ([1,77,768] tensor) or set `pipe.negative_embeddings` "
                                 "(the reference np.load()s negative.npy from a hard-coded path)")
            neg = neg.to(device).float().reshape(-1, seq_len, emb.shape[-1])
            if neg.shape[0] == 1:
                neg = neg.expand(emb.shape[0], -1, -1)
            emb = torch.cat([neg, emb])                                                           # :172
        return emb

    def decode_latents(self, latents):
        """:175-184 -- returns numpy ``[B,3,F,H,W]`` float32 in [0,1] like the reference."""
        video = self.vae.engine.vae_decode(latents, postprocess=True)
        return video.cpu().float().numpy()

    def prepare_extra_step_kwargs(self, generator, eta):                                         # :186-201
        params = set(inspect.signature(self.scheduler.step).parameters.keys())
        kw = {}
        if "eta" in params:
            kw["eta"] = eta
        if "generator" in params:
            kw["generator"] = generator
        return kw

    def check_inputs(self, eeg, height, width, callback_steps):                                  # :203-216
        if not isinstance(eeg, torch.Tensor):
            raise ValueError(f"`eeg` has to be of type `torch.Tensor` but is {type(eeg)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if (callback_steps is None) or (
            callback_steps is not None and (not isinstance(callback_steps, int) or callback_steps <= 0)
        ):
            raise ValueError(
                f"`callback_steps` has to be a positive integer but is {callback_steps} of type"
                f" {type(callback_steps)}."
            )

    def prepare_latents(self, batch_size, num_channels_latents, video_length, height, width, dtype, device, generator,
                        latents=None):                                                           # :218-245
        shape = (batch_size, num_channels_latents, video_length, height // self.vae_scale_factor,
                 width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(
                f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                f" size of {batch_size}. Make sure the batch size matches the length of the generators."
            )
        if latents is None:
            if isinstance(generator, list):
                one = (1,) + shape[1:]
                latents = torch.cat([torch.randn(one, generator=generator[i], device=generator[i].device, dtype=dtype)
                                     for i in range(batch_size)], dim=0).to(device)
            else:
                gdev = generator.device if generator is not None else device
                latents = torch.randn(shape, generator=generator, device=gdev, dtype=dtype).to(device)
        else:
            if tuple(latents.shape) != shape:
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    @torch.no_grad()
    def __call__(
        self,
        model,
        eeg: torch.FloatTensor,
        video_length: Optional[int],
        height: Optional[int] = None,
        width: Optional[int] = None,
        num_inference_steps: int = 50,
        guidance_scale: float = 7.5,
        negative_prompt=None,
        num_videos_per_eeg: Optional[int] = 1,
        eta: float = 0.0,
        generator: Optional[Union[torch.Generator, List[torch.Generator]]] = None,
        latents: Optional[torch.FloatTensor] = None,
        output_type: Optional[str] = "tensor",
        return_dict: bool = True,
        callback: Optional[Callable[[int, int, torch.FloatTensor], None]] = None,
        callback_steps: Optional[int] = 1,
        **kwargs,
    ):
        height = height or self.unet.config.sample_size * self.vae_scale_factor                   # :269-270
        width = width or self.unet.config.sample_size * self.vae_scale_factor
        self.check_inputs(eeg, height, width, callback_steps)                                     # :273
        batch_size = eeg.shape[0]                                                                 # :276
        device = self._execution_device
        do_cfg = guidance_scale > 1.0                                                             # :281
        emb = self._encode_eeg(model, eeg, device, num_videos_per_eeg, do_cfg, negative_prompt)   # :284
        self.scheduler.set_timesteps(num_inference_steps, device=device)                          # :287-288
        timesteps = self.scheduler.timesteps
        latents = self.prepare_latents(batch_size * num_videos_per_eeg, self.unet.in_channels, video_length, height,
                                       width, torch.float32, device, generator, latents)          # :291-302
        if getattr(self.scheduler, "engine", None) is None:      # a scheduler swapped in after construction (pipe.scheduler = ...)
            self.scheduler.bind(self.unet.engine)
        extra = self.prepare_extra_step_kwargs(generator, eta)                                    # :306
        b = latents.shape[0]
        eng = self.unet.engine
        # the fused device loop is the deterministic DDIM one; a stochastic update (eta > 0 draws torch noise per step), another
        # scheduler type or a per-step callback step through the public entry points exactly as the reference's loop does
        fused = (callback is None and self.vae.engine is eng and isinstance(self.scheduler, DDIMScheduler)
                 and extra.get("eta", 0.0) == 0.0)
        if fused:
            self.scheduler.bind(eng)      # the fused loop walks the ctx's schedule: make it this scheduler's (table, steps_offset)
            with self.progress_bar(total=num_inference_steps) as bar:
                video = eng.generate(latents, emb[b:] if do_cfg else emb, emb[:b] if do_cfg else None,
                                     num_inference_steps, guidance_scale, 0.0, decode=True)
                bar.update(num_inference_steps)
            video = video.cpu().float().numpy()                                                   # :183
        else:
            with self.progress_bar(total=num_inference_steps) as bar:
                for i, t in enumerate(timesteps):                                                 # :311
                    x_in = torch.cat([latents] * 2) if do_cfg else latents                        # :313
                    x_in = self.scheduler.scale_model_input(x_in, t)                              # :314
                    eps = self.unet(x_in, t, encoder_hidden_states=emb).sample                    # :317
                    if do_cfg and isinstance(self.scheduler, DDIMScheduler) and extra.get("eta", 0.0) == 0.0:   # :320-325, one kernel
                        eu, ec = eps.chunk(2)
                        latents = eng.ddim_cfg_step(eu, ec, latents, guidance_scale, int(t),
                                                    self.scheduler.prev_timestep(int(t)))
                    elif do_cfg:                                                                  # :320-322 then :325
                        eu, ec = eps.chunk(2)
                        eps = eng.cfg_combine(eu, ec, guidance_scale)
                        latents = self.scheduler.step(eps, t, latents, **extra).prev_sample
                    else:
                        latents = self.scheduler.step(eps, t, latents, **extra).prev_sample
                    bar.update()                                                                  # :328-331
                    if callback is not None and i % callback_steps == 0:
                        callback(i, t, latents)
            video = self.decode_latents(latents)                                                  # :334
        if output_type == "tensor":                                                               # :337-338
            video = torch.from_numpy(video)
        if not return_dict:
            return video
        return TuneAVideoPipelineOutput(videos=video)


def build_pipeline(unet_cfg=None, vae_cfg=None, device: int = 0, seed: int = 42, init: str = "reference_init",
                   unet_sd=None, vae_sd=None) -> TuneAVideoPipeline:
    """One engine holding UNet + VAE (so that the fused loop is used), with given or synthetic weights."""
    from .engine import Engine
    from .weights import UNetConfig, VAEConfig
    unet_cfg, vae_cfg = unet_cfg or UNetConfig(), vae_cfg or VAEConfig()
    eng = Engine(unet_cfg, vae_cfg, device)
    unet = UNet3DConditionModel(sample_size=unet_cfg.sample_size, in_channels=unet_cfg.in_channels,
                                out_channels=unet_cfg.out_channels, block_out_channels=unet_cfg.block_out_channels,
                                layers_per_block=unet_cfg.layers_per_block, cross_attention_dim=unet_cfg.cross_attention_dim,
                                attention_head_dim=unet_cfg.attention_head_dim, norm_num_groups=unet_cfg.norm_num_groups,
                                norm_eps=unet_cfg.norm_eps, engine=eng)
    vae = AutoencoderKL(vae_cfg, engine=eng)
    if unet_sd is not None:
        unet.load_state_dict(unet_sd)
    else:
        unet.init_synthetic(seed, init)
    if vae_sd is not None:
        vae.load_state_dict(vae_sd)
    else:
        vae.init_synthetic(seed + 1, init)
    return TuneAVideoPipeline(vae=vae, tokenizer=None, unet=unet, scheduler=DDIMScheduler(engine=eng))
