"""eeg2video_amd -- the EEG2Video generation hot path (Tune-A-Video denoising loop + SD VAE) on MI355X.

Layout (only what the path needs):

* ``csrc/``      hand-written gfx950 kernels + the C-ABI library ``lib/libeeg2video_hip.so``
* ``_lib`` / ``engine``   ctypes binding and the one-ctx-per-GPU wrapper
* ``unet`` / ``vae`` / ``scheduler`` / ``pipeline``   mirrors of the reference's interfaces
  (``UNet3DConditionModel.forward``, ``AutoencoderKL``, ``DDIMScheduler``, ``TuneAVideoPipeline.__call__``)
* ``weights``    state-dict key scheme + counter-RNG synthetic weights
* ``dist``       sharding of clips over ranks and the all-gather of decoded frames

The compute path has no CPU fallback: without the built library or without a GPU it raises.
"""
from .weights import TINY_SEMANTIC, TINY_UNET, TINY_VAE, SemanticConfig, UNetConfig, VAEConfig  # noqa: F401

__all__ = ["UNetConfig", "VAEConfig", "TINY_UNET", "TINY_VAE", "Engine", "UNet3DConditionModel", "AutoencoderKL",
           "DDIMScheduler", "PNDMScheduler", "TuneAVideoPipeline", "build_pipeline"]


def __getattr__(name):          # lazy: importing the package must not need torch.cuda or the .so
    if name == "Engine":
        from .engine import Engine
        return Engine
    if name == "UNet3DConditionModel":
        from .unet import UNet3DConditionModel
        return UNet3DConditionModel
    if name == "AutoencoderKL":
        from .vae import AutoencoderKL
        return AutoencoderKL
    if name == "DDIMScheduler":
        from .scheduler import DDIMScheduler
        return DDIMScheduler
    if name == "PNDMScheduler":
        from .scheduler import PNDMScheduler
        return PNDMScheduler
    if name in ("TuneAVideoPipeline", "build_pipeline"):
        from . import pipeline
        return getattr(pipeline, name)
    raise AttributeError(name)
