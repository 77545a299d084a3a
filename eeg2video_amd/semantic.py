"""Semantic Predictor ``CLIP`` (the EEG -> CLIP-text-embedding MLP) with the reference's interface, on device.

Mirror of ``EEG2Video/models/train_semantic_predictor.py:11-32`` (same class name, ``forward(eeg)`` returning
``[B, 77*768]``, state-dict keys ``mlp.{0,2,4,6,8}.{weight,bias}``).  This is the step immediately before the hot
path (``pipeline_tuneeeg2video.py:149``); it is a callable, so it drops into ``pipe(model, eeg, ...)`` as ``model``.
"""
from __future__ import annotations

from typing import Optional

import torch

from .engine import Engine
from .weights import SemanticConfig, synth_state_dict, semantic_param_spec


class CLIP:
    def __init__(self, config: SemanticConfig = SemanticConfig(), *, engine: Optional[Engine] = None, device: int = 0):
        self.scfg = config
        self.engine = engine if engine is not None else Engine(device=device, sem_cfg=config)
        if self.engine.sem_cfg != config:
            raise ValueError("the engine was created for a different semantic-predictor config")

    def state_dict_spec(self):
        return semantic_param_spec(self.scfg, self.engine.unet_cfg.cross_attention_dim)

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = state_dict.get("state_dict", state_dict) if isinstance(state_dict, dict) else state_dict   # :146-147 saves {'state_dict': ...}
        spec = self.state_dict_spec()
        missing = [k for k in spec if k not in sd]
        if strict and missing:
            raise RuntimeError(f"Error(s) in loading state_dict for CLIP: missing {missing[:4]}...")
        self.engine.load_state_dict({k: v for k, v in sd.items() if k in spec}, prefix="semantic.")
        self.engine.finalize(Engine.SEMANTIC)
        return self

    def init_synthetic(self, seed: int = 44, mode: str = "reference_init"):
        return self.load_state_dict(synth_state_dict(self.state_dict_spec(), seed=seed, mode=mode))

    def cuda(self):
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def forward(self, eeg: torch.Tensor) -> torch.Tensor:
        return self.engine.semantic_predict(eeg)

    __call__ = forward
