"""eeg2video_amd -- the EEG2Video generation hot path (Tune-A-Video denoising loop + SD VAE) on MI355X.

Layout (only what the path needs):

* ``csrc/``      hand-written gfx950 kernels + the C-ABI library ``lib/libeeg2video_hip.so``
* ``_lib`` / ``engine``   ctypes binding and the one-ctx-per-GPU wrapper
* ``unet`` / ``vae`` / ``scheduler`` / ``pipeline``   mirrors of the reference's interfaces
  (``UNet3DConditionModel.forward`` / ``.from_pretrained``, ``AutoencoderKL``, the six scheduler types the pipeline's
  constructor accepts, ``TuneAVideoPipeline.__call__`` / ``.from_pretrained``)
* ``semantic`` / ``util``   the steps either side of the path: Semantic Predictor, DDIM inversion, ``save_videos_grid``
* ``host_models``   GLMNet / Seq2Seq as plain host-side torch modules (BASELINE configs[4] driver: ``examples/run_sweep.py``)
* ``weights``    state-dict key scheme + counter-RNG synthetic weights
* ``dist``       sharding of clips over ranks and the all-gather of decoded frames

The compute path has no CPU fallback: without the built library or without a GPU it raises.
"""
from .weights import TINY_SEMANTIC, TINY_UNET, TINY_VAE, SemanticConfig, UNetConfig, VAEConfig  # noqa: F401

__all__ = ["UNetConfig", "VAEConfig", "TINY_UNET", "TINY_VAE", "Engine", "UNet3DConditionModel", "AutoencoderKL",
           "DDIMScheduler", "PNDMScheduler", "LMSDiscreteScheduler", "EulerDiscreteScheduler", "EulerAncestralDiscreteScheduler",
           "DPMSolverMultistepScheduler", "TuneAVideoPipeline", "build_pipeline", "save_videos_grid"]


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
    if name in ("DDIMScheduler", "PNDMScheduler", "LMSDiscreteScheduler", "EulerDiscreteScheduler",
                "EulerAncestralDiscreteScheduler", "DPMSolverMultistepScheduler"):
        from . import scheduler
        return getattr(scheduler, name)
    if name == "save_videos_grid":
        from .util import save_videos_grid
        return save_videos_grid
    if name in ("TuneAVideoPipeline", "build_pipeline"):
        from . import pipeline
        return getattr(pipeline, name)
    raise AttributeError(name)
