"""CPU oracle for the EEG2Video generation hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this package, and there only as the checker / the timed CPU baseline.  The
product path (``eeg2video_amd``) never imports it and has no CPU fallback.

What it is: a plain ``torch`` (CPU, fp32) restatement of the reference's algorithm for
the path named by BASELINE.json -- same op sequence as the reference's own CPU path
(``F.conv2d``, ``F.group_norm``, unfused ``baddbmm -> softmax -> bmm`` attention, erf-GELU
GEGLU, DDIM eta = 0).  Every function cites the reference ``file:line`` it follows.

Pinning status (details in DESIGN.md "Oracle"):

* reference-owned code (``resnet.py``, ``attention.py``, ``unet_blocks.py``, ``unet.py``,
  the loop of ``pipeline_tuneeeg2video.py``): PINNED by golden vectors generated in the
  build container from the reference itself (``tests/golden/make_golden.py``): ``resnet.py``
  by direct import, the other three by executing them unmodified with a test-only stand-in
  for the missing ``diffusers`` package.
* arithmetic owned by the absent third-party dependency ``diffusers==0.11.1``
  (``CrossAttention``, ``FeedForward``/``GEGLU``, ``Timesteps``/``TimestepEmbedding``,
  ``AutoencoderKL``, ``DDIMScheduler``): restated from the published algorithm;
  PARITY UNPINNED by the reference (it has no tests or fixtures), anchored only on closed
  forms (timestep lists, alpha-bar table, ``negative.npy`` shape/dtype) and invariants.
"""
from .ddim import DDIMOracle, ddim_loop, next_step  # noqa: F401
from .unet3d import unet3d_forward  # noqa: F401
from .vae import vae_decode, vae_encode  # noqa: F401
from .pipeline import generate  # noqa: F401
from .pndm import PNDMOracle  # noqa: F401
from .schedulers import DPMSolverPPOracle, EulerAncestralOracle, EulerOracle, LMSOracle, lms_coefficient_exact  # noqa: F401
from .extras import dana_noise, frames_to_uint8, semantic_predictor  # noqa: F401
