"""State-dict key scheme of the hot path and deterministic synthetic weights.

The hot path has two weight sets, keyed exactly as the reference's checkpoints are:

* ``UNet3DConditionModel`` -- module attribute names of
  ``EEG2Video/models/unet.py:85-207``, ``unet_blocks.py:196-197,269-281,464-470``,
  ``resnet.py:141-172``, ``attention.py:58-87,158-202`` plus the diffusers 0.11.1
  members they instantiate (``to_q/to_k/to_v/to_out.0``, ``ff.net.0.proj``,
  ``ff.net.2``, ``time_embedding.linear_{1,2}``) -- SURVEY.md App. D.
* ``AutoencoderKL`` (diffusers 0.11.1, not in the reference tree).

There are no checkpoints in the container and no network, so every parity test and
the benchmark use weights drawn from a self-contained counter RNG (SplitMix64 keyed
by ``seed`` and an FNV-1a hash of the tensor name).  It is pure integer numpy, so the
GPU box regenerates bit-identical tensors without any torch-version coupling.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Tuple, Union

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


# --------------------------------------------------------------------------------------
# counter RNG
# --------------------------------------------------------------------------------------
def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """One SplitMix64 output per 64-bit counter value (vectorised, wraps mod 2^64)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream_base(seed: int, name: str) -> np.uint64:
    s = (int(seed) * 0x9E3779B97F4A7C15 + _fnv1a64(name)) & 0xFFFFFFFFFFFFFFFF
    return _splitmix64(np.array([s], dtype=np.uint64))[0]


def counter_uniform(seed: int, name: str, n: int) -> np.ndarray:
    """``n`` float32 values in [0, 1) with 24 random bits each (exact in float32)."""
    base = _stream_base(seed, name)
    with np.errstate(over="ignore"):
        ctr = base + np.arange(n, dtype=np.uint64)
    bits = _splitmix64(ctr) >> np.uint64(40)
    return (bits.astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32)


def counter_normal(seed: int, name: str, shape: Iterable[int]) -> np.ndarray:
    """Standard normals by Box-Muller over two independent counter streams (float32)."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if shape else 1
    u1 = counter_uniform(seed, name + "#u1", n).astype(np.float64)
    u2 = counter_uniform(seed, name + "#u2", n).astype(np.float64)
    u1 = (u1 * 16777216.0 + 1.0) / 16777217.0          # (0, 1): log() is finite
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return z.astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------
# configs
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class UNetConfig:
    """Mirror of the ctor kwargs of ``UNet3DConditionModel`` that the path uses
    (``EEG2Video/models/unet.py:41-78``); defaults are the SD-v1-4 values."""
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    attention_head_dim: Union[int, Tuple[int, ...]] = 8   # = NUMBER of heads (unet_blocks.py:257-259); a tuple: per down block (unet.py:110-111)
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    down_block_types: Tuple[str, ...] = (
        "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D")
    up_block_types: Tuple[str, ...] = (
        "UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D")
    flip_sin_to_cos: bool = True
    freq_shift: int = 0

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4


@dataclass(frozen=True)
class VAEConfig:
    """SD-v1-4 ``vae/config.json`` values."""
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215


@dataclass(frozen=True)
class SemanticConfig:
    """Semantic Predictor MLP ``CLIP`` (``EEG2Video/models/train_semantic_predictor.py:11-32``):
    310 -> 10000 -> 10000 -> 10000 -> 10000 -> 77*768 with ReLU between."""
    in_features: int = 310
    hidden: int = 10000
    tokens: int = 77


TINY_SEMANTIC = SemanticConfig(in_features=22, hidden=96, tokens=7)


def semantic_param_spec(cfg: SemanticConfig, cross_attention_dim: int) -> "OrderedDict[str, Tuple[int, ...]]":
    """``nn.Sequential`` keys of the reference module: ``mlp.{0,2,4,6,8}.{weight,bias}``."""
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    dims = [cfg.in_features, cfg.hidden, cfg.hidden, cfg.hidden, cfg.hidden, cfg.tokens * cross_attention_dim]
    for i in range(5):
        _lin(spec, f"mlp.{2 * i}", dims[i + 1], dims[i])
    return spec


#: tiny configs used by the parity tests (oracle finishes in well under a second).
TINY_UNET = UNetConfig(sample_size=8, block_out_channels=(64, 128, 256, 256), cross_attention_dim=64)
TINY_VAE = VAEConfig(block_out_channels=(32, 64, 64, 64), norm_num_groups=8)


# --------------------------------------------------------------------------------------
# key / shape specs
# --------------------------------------------------------------------------------------
def _conv(spec, name, cout, cin, k):
    spec[name + ".weight"] = (cout, cin, k, k)
    spec[name + ".bias"] = (cout,)


def _lin(spec, name, cout, cin, bias=True):
    spec[name + ".weight"] = (cout, cin)
    if bias:
        spec[name + ".bias"] = (cout,)


def _norm(spec, name, c):
    spec[name + ".weight"] = (c,)
    spec[name + ".bias"] = (c,)


def _resnet3d(spec, p, cin, cout, temb):
    # resnet.py:141-172
    _norm(spec, p + ".norm1", cin)
    _conv(spec, p + ".conv1", cout, cin, 3)
    _lin(spec, p + ".time_emb_proj", cout, temb)
    _norm(spec, p + ".norm2", cout)
    _conv(spec, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(spec, p + ".conv_shortcut", cout, cin, 1)


def _transformer3d(spec, p, c, cross):
    # attention.py:58-87 (norm, proj_in, proj_out) and :158-202 (block members)
    _norm(spec, p + ".norm", c)
    _conv(spec, p + ".proj_in", c, c, 1)
    b = p + ".transformer_blocks.0"
    for a, kv in (("attn1", c), ("attn2", cross), ("attn_temp", c)):
        _lin(spec, f"{b}.{a}.to_q", c, c, bias=False)
        _lin(spec, f"{b}.{a}.to_k", c, kv, bias=False)
        _lin(spec, f"{b}.{a}.to_v", c, kv, bias=False)
        _lin(spec, f"{b}.{a}.to_out.0", c, c)
    for n in ("norm1", "norm2", "norm3", "norm_temp"):
        _norm(spec, f"{b}.{n}", c)
    _lin(spec, f"{b}.ff.net.0.proj", 8 * c, c)
    _lin(spec, f"{b}.ff.net.2", c, 4 * c)
    _conv(spec, p + ".proj_out", c, c, 1)


def unet_param_spec(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape for every tensor of the reference UNet state dict (App. D)."""
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = cfg.block_out_channels
    temb = cfg.time_embed_dim
    _conv(spec, "conv_in", boc[0], cfg.in_channels, 3)                  # unet.py:85
    _lin(spec, "time_embedding.linear_1", temb, boc[0])                 # unet.py:91
    _lin(spec, "time_embedding.linear_2", temb, temb)
    out_c = boc[0]
    for i, typ in enumerate(cfg.down_block_types):                      # unet.py:113-139
        in_c, out_c = out_c, boc[i]
        for j in range(cfg.layers_per_block):
            _resnet3d(spec, f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c, temb)
            if typ == "CrossAttnDownBlock3D":
                _transformer3d(spec, f"down_blocks.{i}.attentions.{j}", out_c, cfg.cross_attention_dim)
        if i != len(boc) - 1:
            _conv(spec, f"down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)
    mid = boc[-1]                                                       # unet.py:142-156
    _resnet3d(spec, "mid_block.resnets.0", mid, mid, temb)
    _transformer3d(spec, "mid_block.attentions.0", mid, cfg.cross_attention_dim)
    _resnet3d(spec, "mid_block.resnets.1", mid, mid, temb)
    rev = list(reversed(boc))                                           # unet.py:164-202
    out_c = rev[0]
    for i, typ in enumerate(cfg.up_block_types):
        prev = out_c
        out_c = rev[i]
        in_c = rev[min(i + 1, len(boc) - 1)]
        n_layers = cfg.layers_per_block + 1
        for j in range(n_layers):
            skip = in_c if j == n_layers - 1 else out_c                 # unet_blocks.py:431-432
            rin = prev if j == 0 else out_c
            _resnet3d(spec, f"up_blocks.{i}.resnets.{j}", rin + skip, out_c, temb)
            if typ == "CrossAttnUpBlock3D":
                _transformer3d(spec, f"up_blocks.{i}.attentions.{j}", out_c, cfg.cross_attention_dim)
        if i != len(boc) - 1:
            _conv(spec, f"up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
    _norm(spec, "conv_norm_out", boc[0])                                # unet.py:205-207
    _conv(spec, "conv_out", cfg.out_channels, boc[0], 3)
    return spec


def _resnet2d(spec, p, cin, cout):
    _norm(spec, p + ".norm1", cin)
    _conv(spec, p + ".conv1", cout, cin, 3)
    _norm(spec, p + ".norm2", cout)
    _conv(spec, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(spec, p + ".conv_shortcut", cout, cin, 1)


def _vae_mid(spec, p, c):
    _resnet2d(spec, p + ".resnets.0", c, c)
    a = p + ".attentions.0"
    _norm(spec, a + ".group_norm", c)
    for n in ("query", "key", "value", "proj_attn"):
        _lin(spec, f"{a}.{n}", c, c)
    _resnet2d(spec, p + ".resnets.1", c, c)


def vae_param_spec(cfg: VAEConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape of the diffusers-0.11.1 ``AutoencoderKL`` state dict (App. C.5)."""
    spec: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = cfg.block_out_channels
    # encoder
    _conv(spec, "encoder.conv_in", boc[0], cfg.in_channels, 3)
    out_c = boc[0]
    for i in range(len(boc)):
        in_c, out_c = out_c, boc[i]
        for j in range(cfg.layers_per_block):
            _resnet2d(spec, f"encoder.down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
        if i != len(boc) - 1:
            _conv(spec, f"encoder.down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)
    _vae_mid(spec, "encoder.mid_block", boc[-1])
    _norm(spec, "encoder.conv_norm_out", boc[-1])
    _conv(spec, "encoder.conv_out", 2 * cfg.latent_channels, boc[-1], 3)
    # decoder
    rev = list(reversed(boc))
    _conv(spec, "decoder.conv_in", rev[0], cfg.latent_channels, 3)
    _vae_mid(spec, "decoder.mid_block", rev[0])
    out_c = rev[0]
    for i in range(len(boc)):
        in_c, out_c = out_c, rev[i]
        for j in range(cfg.layers_per_block + 1):
            _resnet2d(spec, f"decoder.up_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
        if i != len(boc) - 1:
            _conv(spec, f"decoder.up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)
    _norm(spec, "decoder.conv_norm_out", boc[0])
    _conv(spec, "decoder.conv_out", cfg.out_channels, boc[0], 3)
    _conv(spec, "quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    _conv(spec, "post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    return spec


# --------------------------------------------------------------------------------------
# synthetic tensors
# --------------------------------------------------------------------------------------
def _is_norm(name: str) -> bool:
    leaf = name.rsplit(".", 2)[-2]
    return leaf.startswith("norm") or leaf in ("group_norm", "conv_norm_out")


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int = 42, mode: str = "perturbed",
                 fan_in: int = 0) -> np.ndarray:
    """One synthetic parameter.

    ``mode="reference_init"``: torch-default-style init (``U(+-1/sqrt(fan_in))`` for conv and
    linear weight and bias; norm gamma=1, beta=0; ``attn_temp.to_out.0.weight`` = 0 as
    ``attention.py:201`` does) -- BASELINE config 1.
    ``mode="perturbed"``: same, but norm affine parameters are ``1 + 0.2u`` / ``0.2u`` and the
    temporal ``to_out`` weight is non-zero, so that no term of the path is hidden by a
    neutral parameter in a parity test.
    """
    n = int(np.prod(shape))
    if _is_norm(name):
        if mode == "reference_init":
            v = np.ones(n, np.float32) if name.endswith(".weight") else np.zeros(n, np.float32)
        else:
            u = counter_uniform(seed, name, n) * 2.0 - 1.0
            v = (1.0 + 0.2 * u if name.endswith(".weight") else 0.2 * u).astype(np.float32)
        return v.reshape(shape)
    if mode == "reference_init" and name.endswith("attn_temp.to_out.0.weight"):
        return np.zeros(shape, np.float32)
    if name.endswith(".weight"):
        fan_in = int(np.prod(shape[1:]))
    elif fan_in <= 0:                      # bias without a known sibling weight
        fan_in = shape[0]
    bound = 1.0 / math.sqrt(max(fan_in, 1))
    u = counter_uniform(seed, name, n) * 2.0 - 1.0
    return (u * bound).astype(np.float32).reshape(shape)


def synth_state_dict(spec: "OrderedDict[str, Tuple[int, ...]]", seed: int = 42,
                     mode: str = "perturbed") -> "OrderedDict[str, np.ndarray]":
    """All tensors of ``spec`` (numpy float32), bias bounds taken from the sibling weight."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in spec.items():
        fan_in = 0
        if name.endswith(".bias"):
            w = spec.get(name[:-5] + ".weight")
            if w is not None and len(w) > 1:
                fan_in = int(np.prod(w[1:]))
        out[name] = synth_tensor(name, shape, seed, mode, fan_in)
    return out


def synth_inputs(batch: int, cfg: UNetConfig, frames: int, h: int, w: int, n_tokens: int = 77,
                 seed: int = 1234):
    """BASELINE config-1 style inputs: latents ``[B,4,F,h,w]`` (seed+k per clip), cond and
    uncond ``[B,77,cross]`` / ``[1,77,cross]`` from their own seeds."""
    lat = np.stack([counter_normal(seed + k, "latent", (cfg.in_channels, frames, h, w)) for k in range(batch)])
    cond = np.stack([counter_normal(seed + 1 + 7919 * k, "cond", (n_tokens, cfg.cross_attention_dim))
                     for k in range(batch)])
    uncond = counter_normal(seed + 2, "uncond", (1, n_tokens, cfg.cross_attention_dim))
    return lat, cond, uncond
