"""``AutoencoderKL`` look-alike over the HIP library.

The VAE belongs to diffusers 0.11.1 (not in the reference tree); this mirrors the members the
reference touches: ``decode(z).sample`` (``pipeline_tuneeeg2video.py:179``), ``encode(x).latent_dist``
with ``.sample()`` / ``.mean`` (``train_finetune_videodiffusion.py:264``, ``generate_1200_latent.py:38``),
``config.block_out_channels`` (``pipeline:113``), ``enable_slicing`` / ``disable_slicing`` (:115-119).
"""
from __future__ import annotations

import json
import os
from typing import Optional

import torch

from .engine import Engine
from .unet import FrozenDict
from .weights import UNetConfig, VAEConfig, synth_state_dict, vae_param_spec


class DecoderOutput:
    def __init__(self, sample):
        self.sample = sample

    def __getitem__(self, k):
        return self.sample if k in ("sample", 0) else (_ for _ in ()).throw(KeyError(k))


class DiagonalGaussianDistribution:
    """mean / logvar (clamped to [-30, 20]) as diffusers builds them from the encoder moments."""

    def __init__(self, mean: torch.Tensor, logvar: torch.Tensor):
        self.mean, self.logvar = mean, logvar
        self.std = torch.exp(0.5 * logvar)
        self.var = torch.exp(logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKLOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class AutoencoderKL:
    def __init__(self, config: VAEConfig = VAEConfig(), *, engine: Optional[Engine] = None,
                 unet_config: Optional[UNetConfig] = None, device: int = 0):
        self.vcfg = config
        self._internal_dict = FrozenDict(block_out_channels=tuple(config.block_out_channels),
                                         latent_channels=config.latent_channels, in_channels=config.in_channels,
                                         out_channels=config.out_channels, layers_per_block=config.layers_per_block,
                                         norm_num_groups=config.norm_num_groups)
        self.engine = engine if engine is not None else Engine(unet_config or UNetConfig(), config, device)
        self.use_slicing = False

    @property
    def config(self):
        return self._internal_dict

    @property
    def dtype(self):
        return torch.float32

    @property
    def device(self):
        return self.engine.device

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def enable_slicing(self):      # the library always decodes clip by clip; the flag is kept for drop-in use
        self.use_slicing = True

    def disable_slicing(self):
        self.use_slicing = False

    def state_dict_spec(self):
        return vae_param_spec(self.vcfg)

    def load_state_dict(self, state_dict, strict: bool = True):
        spec = self.state_dict_spec()
        missing = [k for k in spec if k not in state_dict]
        if strict and missing:
            raise RuntimeError(f"Error(s) in loading state_dict for AutoencoderKL: missing {missing[:4]}...")
        self.engine.load_state_dict({k: v for k, v in state_dict.items() if k in spec}, prefix="vae.")
        self.engine.finalize(Engine.VAE)
        return self

    def init_synthetic(self, seed: int = 43, mode: str = "reference_init"):
        return self.load_state_dict(synth_state_dict(self.state_dict_spec(), seed=seed, mode=mode))

    @staticmethod
    def config_from_dir(path: str) -> VAEConfig:
        """``config.json`` of a diffusers ``AutoencoderKL`` -> ``VAEConfig`` (0.11.1 configs carry no ``scaling_factor``: 0.18215)."""
        cfg_file = os.path.join(path, "config.json")
        if not os.path.isfile(cfg_file):
            raise RuntimeError(f"{cfg_file} does not exist")
        with open(cfg_file) as f:
            cj = json.load(f)
        if cj.get("down_block_types") and any(t != "DownEncoderBlock2D" for t in cj["down_block_types"]):
            raise NotImplementedError(f"VAE block types {cj['down_block_types']} are not the Stable-Diffusion ones")
        return VAEConfig(in_channels=cj.get("in_channels", 3), out_channels=cj.get("out_channels", 3),
                         latent_channels=cj.get("latent_channels", 4),
                         block_out_channels=tuple(cj.get("block_out_channels", (128, 256, 512, 512))),
                         layers_per_block=cj.get("layers_per_block", 2), norm_num_groups=cj.get("norm_num_groups", 32),
                         scaling_factor=cj.get("scaling_factor", 0.18215))

    @classmethod
    def from_pretrained(cls, pretrained_model_path: str, subfolder: Optional[str] = None, torch_dtype=None, *,
                        engine: Optional[Engine] = None, unet_config: Optional[UNetConfig] = None, device: int = 0):
        """Local directory only (what ``DiffusionPipeline.from_pretrained`` does for the ``vae`` component of
        ``inference_eeg2video.py:70``): ``config.json`` (diffusers ``AutoencoderKL`` config) + ``diffusion_pytorch_model.bin``
        (or ``.safetensors``).  ``torch_dtype`` is accepted for drop-in use; fp16 / bf16 checkpoints are widened to fp32 at load."""
        path = os.path.join(pretrained_model_path, subfolder) if subfolder else pretrained_model_path
        config = cls.config_from_dir(path)
        if engine is not None and engine.vae_cfg != config:
            raise RuntimeError(f"the engine was created for a VAE of {engine.vae_cfg}, the checkpoint holds {config}: create the UNet "
                               "with vae_config=AutoencoderKL.config_from_dir(<dir>/vae)")
        model = cls(config, engine=engine, unet_config=unet_config, device=device)
        from .unet import SAFETENSORS_NAME, WEIGHTS_NAME, _load_checkpoint
        if not os.path.isfile(os.path.join(path, WEIGHTS_NAME)) and not os.path.isfile(os.path.join(path, SAFETENSORS_NAME)):
            raise RuntimeError(f"{os.path.join(path, WEIGHTS_NAME)} does not exist")
        sd = _load_checkpoint(path)          # .safetensors preferred; .bin with weights_only=True
        return model.load_state_dict(sd)

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        out = self.engine.vae_decode(z, postprocess=False)
        return DecoderOutput(out) if return_dict else (out,)

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        mean, logvar = self.engine.vae_encode(x)
        dist = DiagonalGaussianDistribution(mean, logvar)
        return AutoencoderKLOutput(dist) if return_dict else (dist,)
