"""Host-side EEG models of BASELINE configs[4] -- plain PyTorch modules, kept OUTSIDE the HIP path on purpose (north_star:
"the GLMNet EEG encoder and Seq2Seq latent predictor stay as lightweight PyTorch-ROCm host code").  They exist so that the
end-to-end driver (``examples/run_sweep.py``) can feed the accelerated path with tensors of the right shapes and measure the
host stages' share; weights are random-init (no checkpoints offline), so nothing here is parity-tested against the reference.

Architectures (written from the shapes the reference documents, not copied):
* ``GLMNet``  -- ``glfnet`` of ``EEG2Video/models/models.py:352-375``: a global ShallowNet over all 62 channels and a local one
  over the 12 occipital channels (indices 50..61), concatenated into a linear head.  ``ShallowNet`` (``:105-123``): temporal
  conv (1x25) -> spatial conv (Cx1) -> BatchNorm -> ELU -> average pool (1x51, stride 5) -> linear.
* ``Seq2SeqLatents`` -- ``myTransformer`` of ``EEG2Video_New/Seq2Seq/my_autoregressive_transformer.py:123-192``: EEGNet-style
  embedding of 7 EEG windows ``[B,7,62,100]`` -> 2-layer Transformer encoder -> 4-layer decoder run autoregressively for 6
  steps from a zero token -> linear head to ``[B,7,4,36,64]`` latents (the first is the start token; ``[:, 1:]`` are the 6 frames).
"""
from __future__ import annotations

import math

import torch
from torch import nn


class ShallowNet(nn.Module):
    def __init__(self, out_dim: int, C: int, T: int):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(1, 40, (1, 25)), nn.Conv2d(40, 40, (C, 1)), nn.BatchNorm2d(40), nn.ELU(),
                                 nn.AvgPool2d((1, 51), (1, 5)), nn.Dropout(0.5))
        self.out = nn.Linear(1040 * (T // 200), out_dim)

    def forward(self, x):                       # [B, 1, C, T]
        return self.out(self.net(x).flatten(1))


class GLMNet(nn.Module):
    OCCIPITAL = tuple(range(50, 62))

    def __init__(self, out_dim: int = 2, emb_dim: int = 64, C: int = 62, T: int = 200):
        super().__init__()
        self.globalnet = ShallowNet(emb_dim, C, T)
        self.occipital_localnet = ShallowNet(emb_dim, len(self.OCCIPITAL), T)
        self.out = nn.Linear(2 * emb_dim, out_dim)

    def forward(self, x):                       # [B, 1, C, T]
        g = self.globalnet(x)
        o = self.occipital_localnet(x[:, :, list(self.OCCIPITAL), :])
        return self.out(torch.cat((g, o), 1))


class EEGNetEmbedding(nn.Module):
    """EEGNet (F1 = 16, D = 4, F2 = 16) over one window ``[1, 62, 100]`` -> ``d_model`` (my_autoregressive_transformer.py:16-90)."""

    def __init__(self, d_model: int = 512, C: int = 62, T: int = 100, F1: int = 16, D: int = 4, F2: int = 16):
        super().__init__()
        self.block_1 = nn.Sequential(nn.ZeroPad2d((31, 32, 0, 0)), nn.Conv2d(1, F1, (1, 64), bias=False), nn.BatchNorm2d(F1))
        self.block_2 = nn.Sequential(nn.Conv2d(F1, F1 * D, (C, 1), groups=F1, bias=False), nn.BatchNorm2d(F1 * D), nn.ELU(),
                                     nn.AvgPool2d((1, 4)), nn.Dropout(0.5))
        self.block_3 = nn.Sequential(nn.ZeroPad2d((7, 8, 0, 0)), nn.Conv2d(F1 * D, F1 * D, (1, 16), groups=F1 * D, bias=False),
                                     nn.Conv2d(F1 * D, F2, (1, 1), bias=False), nn.BatchNorm2d(F2), nn.ELU(), nn.AvgPool2d((1, 8)),
                                     nn.Dropout(0.5))
        self.embedding = nn.Linear(F2 * (T // 32), d_model)

    def forward(self, x):                       # [N, 1, C, T]
        return self.embedding(self.block_3(self.block_2(self.block_1(x))).flatten(1))


class PositionalEncoding(nn.Module):
    def __init__(self, d_model: int, max_len: int = 64):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        pos = torch.arange(max_len, dtype=torch.float32)[:, None]
        div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
        pe[:, 0::2], pe[:, 1::2] = torch.sin(pos * div), torch.cos(pos * div)
        self.register_buffer("pe", pe[None])

    def forward(self, x):
        return x + self.pe[:, : x.size(1)]


class Seq2SeqLatents(nn.Module):
    def __init__(self, d_model: int = 512, latent_shape=(4, 36, 64), windows: int = 7, frames: int = 6):
        super().__init__()
        self.latent_shape, self.windows, self.frames = tuple(latent_shape), windows, frames
        n_lat = latent_shape[0] * latent_shape[1] * latent_shape[2]
        self.eeg_embedding = EEGNetEmbedding(d_model)
        self.transformer_encoder = nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model, nhead=4, batch_first=True), num_layers=2)
        self.transformer_decoder = nn.TransformerDecoder(nn.TransformerDecoderLayer(d_model, nhead=4, batch_first=True), num_layers=4)
        self.positional_encoding = PositionalEncoding(d_model)
        self.predictor = nn.Linear(d_model, n_lat)

    @torch.no_grad()
    def forward(self, src):                     # [B, 7, 62, 100] -> latents [B, frames, 4, h, w]
        b = src.shape[0]
        e = self.eeg_embedding(src.reshape(b * self.windows, 1, src.shape[2], src.shape[3])).reshape(b, self.windows, -1)
        mem = self.transformer_encoder(self.positional_encoding(e))
        tgt = torch.zeros(b, 1, e.shape[-1], device=src.device, dtype=e.dtype)
        mask = nn.Transformer.generate_square_subsequent_mask(self.frames + 1).to(src.device)
        for i in range(self.frames):            # autoregressive: the decoder sees the tokens produced so far
            out = self.transformer_decoder(tgt, mem, tgt_mask=mask[: i + 1, : i + 1])
            tgt = torch.cat((tgt, out[:, -1:, :]), dim=1)
        lat = self.predictor(tgt).reshape(b, self.frames + 1, *self.latent_shape)
        return lat[:, 1:]
