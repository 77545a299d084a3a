"""Oracle for the steps either side of the hot path -- TEST INFRASTRUCTURE (see oracle/__init__.py)."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def semantic_predictor(sd, eeg: torch.Tensor) -> torch.Tensor:
    """``CLIP.forward`` -- EEG2Video/models/train_semantic_predictor.py:14-32 (Linear/ReLU x4, Linear)."""
    x = eeg
    for i in range(5):
        x = F.linear(x, sd[f"mlp.{2 * i}.weight"], sd[f"mlp.{2 * i}.bias"])
        if i < 4:
            x = F.relu(x)
    return x


def dana_noise(x0: torch.Tensor, eps_div: torch.Tensor, eps_same: torch.Tensor, t: torch.Tensor, dynamic_beta: float,
               time_steps: int = 500) -> torch.Tensor:
    """``Diffusion.forward`` -- EEG2Video/models/DANA_module.py:52-72 with the random draws passed in
    (``eps_div`` = ``diverse_noise``, ``eps_same`` = ``same_noise_i``), followed by the caller's layout fix
    ``'a b c d e -> a c b d e'`` (EEG2Video/inference_eeg2video.py:77,82)."""
    betas = torch.linspace(0.0001, 0.02, time_steps)                                  # :42-52
    ac = torch.cumprod(1.0 - betas, dim=0)                                            # :15
    b, f = x0.shape[:2]
    diverse = eps_div * math.sqrt(1 - dynamic_beta)                                   # :60
    same = eps_same.repeat(1, f, 1, 1, 1) * math.sqrt(dynamic_beta)                   # :58,61
    a = torch.sqrt(ac)[t].reshape(b, 1, 1, 1, 1)                                      # :63-65
    s = torch.sqrt(1.0 - ac)[t].reshape(b, 1, 1, 1, 1)                                # :67-69
    out = a * x0 + s * (diverse + same)                                               # :71-72
    return out.permute(0, 2, 1, 3, 4).contiguous()


def frames_to_uint8(videos: torch.Tensor) -> np.ndarray:
    """``(x * 255).numpy().astype(np.uint8)`` -- EEG2Video_New/Generation/tuneavideo/util.py:29."""
    return (videos * 255).numpy().astype(np.uint8)
