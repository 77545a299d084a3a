"""Oracle: Stable-Diffusion ``AutoencoderKL`` encode / decode with plain torch CPU ops.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  The VAE lives entirely in the absent
third-party dependency ``diffusers==0.11.1`` (``requirements.txt:4``); this file restates
its published algorithm -- PARITY UNPINNED by the reference.  Reference
call sites: decode ``EEG2Video/pipelines/pipeline_tuneeeg2video.py:175-184``; encode
``EEG2Video/train_finetune_videodiffusion.py:260-267`` and
``EEG2Video_New/Seq2Seq/generate_1200_latent.py:35-38``.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def _resnet2d(sd: SD, p: str, x: torch.Tensor, groups: int, eps: float) -> torch.Tensor:
    """[dep ``ResnetBlock2D``, temb=None, output_scale_factor=1]."""
    h = F.silu(F.group_norm(x, groups, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    h = F.silu(F.group_norm(h, groups, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps))
    h = F.conv2d(h, sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + ".conv_shortcut.weight"], sd[p + ".conv_shortcut.bias"])
    return x + h


def _attention_block(sd: SD, p: str, x: torch.Tensor, groups: int, eps: float) -> torch.Tensor:
    """[dep ``AttentionBlock``: one head, softmax in fp32, rescale_output_factor=1]."""
    b, c, h, w = x.shape
    res = x
    y = F.group_norm(x, groups, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], eps)
    y = y.view(b, c, h * w).transpose(1, 2)
    q = F.linear(y, sd[p + ".query.weight"], sd[p + ".query.bias"])
    k = F.linear(y, sd[p + ".key.weight"], sd[p + ".key.bias"])
    v = F.linear(y, sd[p + ".value.weight"], sd[p + ".value.bias"])
    scale = 1.0 / math.sqrt(c / 1)
    scores = torch.baddbmm(torch.empty(b, h * w, h * w, dtype=q.dtype), q, k.transpose(-1, -2), beta=0, alpha=scale)
    probs = torch.softmax(scores.float(), dim=-1).type(scores.dtype)
    y = torch.bmm(probs, v)
    y = F.linear(y, sd[p + ".proj_attn.weight"], sd[p + ".proj_attn.bias"])
    y = y.transpose(-1, -2).reshape(b, c, h, w)
    return y + res


def _mid(sd: SD, p: str, x: torch.Tensor, groups: int, eps: float) -> torch.Tensor:
    x = _resnet2d(sd, p + ".resnets.0", x, groups, eps)
    x = _attention_block(sd, p + ".attentions.0", x, groups, eps)
    return _resnet2d(sd, p + ".resnets.1", x, groups, eps)


def vae_decode(sd: SD, cfg, z: torch.Tensor) -> torch.Tensor:
    """[dep ``AutoencoderKL.decode``]: ``z [n,4,h,w]`` -> image ``[n,3,8h,8w]`` (no clamp)."""
    g, eps = cfg.norm_num_groups, cfg.norm_eps
    nb = len(cfg.block_out_channels)
    x = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = _mid(sd, "decoder.mid_block", x, g, eps)
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            x = _resnet2d(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, g, eps)
        if i != nb - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
            x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=1)
    x = F.silu(F.group_norm(x, g, sd["decoder.conv_norm_out.weight"], sd["decoder.conv_norm_out.bias"], eps))
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def vae_encode(sd: SD, cfg, img: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """[dep ``AutoencoderKL.encode``]: image ``[n,3,H,W]`` -> (mean, logvar) ``[n,4,H/8,W/8]``;
    ``logvar`` clamped to [-30, 20] as ``DiagonalGaussianDistribution`` does."""
    g, eps = cfg.norm_num_groups, cfg.norm_eps
    nb = len(cfg.block_out_channels)
    x = F.conv2d(img, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            x = _resnet2d(sd, f"encoder.down_blocks.{i}.resnets.{j}", x, g, eps)
        if i != nb - 1:
            p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
            x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=2, padding=0)
    x = _mid(sd, "encoder.mid_block", x, g, eps)
    x = F.silu(F.group_norm(x, g, sd["encoder.conv_norm_out.weight"], sd["encoder.conv_norm_out.bias"], eps))
    x = F.conv2d(x, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    x = F.conv2d(x, sd["quant_conv.weight"], sd["quant_conv.bias"])
    mean, logvar = x.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)
