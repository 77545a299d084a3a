"""``AutoencoderKL`` look-alike over the HIP library.

The VAE belongs to diffusers 0.11.1 (not in the reference tree); this mirrors the members the
reference touches: ``decode(z).sample`` (``pipeline_tuneeeg2video.py:179``), ``encode(x).latent_dist``
with ``.sample()`` / ``.mean`` (``train_finetune_videodiffusion.py:264``, ``generate_1200_latent.py:38``),
``config.block_out_channels`` (``pipeline:113``), ``enable_slicing`` / ``disable_slicing`` (:115-119).
"""
from __future__ import annotations

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

    def decode(self, z: torch.Tensor, return_dict: bool = True):
        out = self.engine.vae_decode(z, postprocess=False)
        return DecoderOutput(out) if return_dict else (out,)

    def encode(self, x: torch.Tensor, return_dict: bool = True):
        mean, logvar = self.engine.vae_encode(x)
        dist = DiagonalGaussianDistribution(mean, logvar)
        return AutoencoderKLOutput(dist) if return_dict else (dist,)
