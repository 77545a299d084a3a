"""``UNet3DConditionModel`` with the reference's interface, executed by the HIP library.

Mirror of ``EEG2Video/models/unet.py:37-449``: same constructor kwargs, ``forward`` signature
(:278-286), output type (attribute and key access, ``tuneavideo/util.py:70``), attributes the
pipeline reads (``in_channels`` ``pipeline_tuneeeg2video.py:291``, ``config.sample_size`` :269,
``dtype``, ``to()``), ``from_pretrained`` / ``from_pretrained_2d`` (:415-449) from a LOCAL directory.
Options of the reference ctor that the SD-v1-4 checkpoint never uses raise ``NotImplementedError``.
"""
from __future__ import annotations

import json
import os
from typing import Optional, Tuple, Union

import numpy as np
import torch

from .engine import Engine
from .weights import UNetConfig, VAEConfig, synth_state_dict, unet_param_spec

WEIGHTS_NAME = "diffusion_pytorch_model.bin"      # diffusers.utils.WEIGHTS_NAME (unet.py:439-441)
SAFETENSORS_NAME = "diffusion_pytorch_model.safetensors"


def _load_checkpoint(path: str) -> dict:
    """State dict of a diffusers-layout model directory: the ``.safetensors`` file when there is one, else the ``.bin`` unpickled
    with ``weights_only=True`` (a state dict of tensors needs nothing more; a checkpoint directory is user-supplied input)."""
    st = os.path.join(path, SAFETENSORS_NAME)
    if os.path.isfile(st):
        from safetensors.torch import load_file
        return dict(load_file(st, device="cpu"))
    return torch.load(os.path.join(path, WEIGHTS_NAME), map_location="cpu", weights_only=True)


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class UNet3DConditionOutput:
    """``BaseOutput`` look-alike (unet.py:32-34): ``out.sample`` and ``out["sample"]`` both work."""

    def __init__(self, sample: torch.Tensor):
        self.sample = sample

    def __getitem__(self, k):
        return self.sample if k in ("sample", 0) else (_ for _ in ()).throw(KeyError(k))

    def to_tuple(self):
        return (self.sample,)


class UNet3DConditionModel:
    def __init__(
        self,
        sample_size: Optional[int] = None,
        in_channels: int = 4,
        out_channels: int = 4,
        center_input_sample: bool = False,
        flip_sin_to_cos: bool = True,
        freq_shift: int = 0,
        down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
        mid_block_type: str = "UNetMidBlock3DCrossAttn",
        up_block_types: Tuple[str, ...] = ("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
        only_cross_attention: Union[bool, Tuple[bool, ...]] = False,
        block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280),
        layers_per_block: int = 2,
        downsample_padding: int = 1,
        mid_block_scale_factor: float = 1,
        act_fn: str = "silu",
        norm_num_groups: int = 32,
        norm_eps: float = 1e-5,
        cross_attention_dim: int = 1280,
        attention_head_dim: Union[int, Tuple[int, ...]] = 8,
        dual_cross_attention: bool = False,
        use_linear_projection: bool = False,
        class_embed_type: Optional[str] = None,
        num_class_embeds: Optional[int] = None,
        upcast_attention: bool = False,
        resnet_time_scale_shift: str = "default",
        *,
        engine: Optional[Engine] = None,
        vae_config: Optional[VAEConfig] = None,
        device: int = 0,
    ):
        cfg = dict(locals())
        for k in ("self", "engine", "vae_config", "device"):
            cfg.pop(k)
        self._internal_dict = FrozenDict(cfg)
        unsupported = {
            "center_input_sample": center_input_sample, "dual_cross_attention": dual_cross_attention,
            "only_cross_attention": only_cross_attention if isinstance(only_cross_attention, bool) else any(only_cross_attention),
        }
        for k, v in unsupported.items():
            if v:
                raise NotImplementedError(f"{k}={v!r} is outside the SD-v1-4 / Tune-A-Video path this build accelerates")
        if class_embed_type is not None or num_class_embeds is not None:
            raise NotImplementedError("class embeddings are not part of the accelerated path")
        if tuple(down_block_types) != UNetConfig.down_block_types or tuple(up_block_types) != UNetConfig.up_block_types:
            raise ValueError(f"unsupported block types {down_block_types} / {up_block_types}")
        if mid_block_type != "UNetMidBlock3DCrossAttn":
            raise ValueError(f"unknown mid_block_type : {mid_block_type}")                       # unet.py:158
        if resnet_time_scale_shift != "default" or act_fn not in ("silu", "swish") or downsample_padding != 1 \
                or mid_block_scale_factor != 1:
            raise NotImplementedError("only time_embedding_norm='default', SiLU, padding 1, scale 1 are implemented")
        if not isinstance(attention_head_dim, int):           # one head count per down block (unet.py:110-111; SD-2.x: 5 / 10 / 20 / 20)
            attention_head_dim = tuple(int(h) for h in attention_head_dim)
            if len(attention_head_dim) != len(block_out_channels):
                raise ValueError(f"attention_head_dim has {len(attention_head_dim)} entries for {len(block_out_channels)} blocks")
            if len(set(attention_head_dim)) == 1:
                attention_head_dim = attention_head_dim[0]
        # Transformer3DModel(use_linear_projection=True) (attention.py:60-63,83-86: Linear-shaped proj_in / proj_out weights, as SD-2.x
        # checkpoints carry them -- with the per-block attention_head_dim (5, 10, 20, 20) above, the SD-2.x UNet shape) applies proj_in / proj_out
        # as nn.Linear on the tokens instead of a 1x1 Conv2d on the map (attention.py:99-123): with channel-last rows the two are the
        # same GEMM, so the option only changes the SHAPE those two weights have in a state dict ([C, C] instead of [C, C, 1, 1])
        self.use_linear_projection = bool(use_linear_projection)
        # upcast_attention=True (attention.py:232-243 -> CrossAttention: q / k widened to fp32 for the scores and the softmax of a half
        # model) asks for what this path always does: scores accumulate and the softmax runs in fp32 in BOTH modes (fp32: everything;
        # bf16: bf16 Q / K operands, fp32 accumulation, fp32 softmax) -- accepted, nothing to switch
        self.upcast_attention = bool(upcast_attention)
        self.sample_size = sample_size
        self.in_channels = in_channels
        self.ucfg = UNetConfig(sample_size=sample_size or 64, in_channels=in_channels, out_channels=out_channels,
                               block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                               cross_attention_dim=cross_attention_dim, attention_head_dim=attention_head_dim,
                               norm_num_groups=norm_num_groups, norm_eps=norm_eps,
                               flip_sin_to_cos=flip_sin_to_cos, freq_shift=freq_shift)
        self.engine = engine if engine is not None else Engine(self.ucfg, vae_config or VAEConfig(), device)
        self.training = False

    # -- attributes the callers rely on -------------------------------------------------------
    @property
    def config(self) -> FrozenDict:
        return self._internal_dict

    @property
    def dtype(self) -> torch.dtype:
        return torch.float32

    @property
    def device(self) -> torch.device:
        return self.engine.device

    def to(self, *args, **kwargs):
        """Weights live on the engine's GPU; a device argument is a no-op.  A floating dtype selects the ARITHMETIC as it does for
        the reference module: ``.to(torch.float16)`` / ``torch_dtype=torch.float16`` (``inference_eeg2video.py:69-70``) = the fp16
        mode, ``torch.bfloat16`` = the bf16 mode, ``torch.float32`` = the fp32 mode."""
        dt = kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dt = a
        if dt is not None and dt.is_floating_point:
            self.engine.set_compute_dtype(dt)
        return self

    def eval(self):
        return self

    def half(self):                         # the reference's `.half()` run: the same kernels on IEEE half
        return self.to(torch.float16)

    def float(self):
        return self.to(torch.float32)

    def requires_grad_(self, flag: bool = False):
        return self

    # -- memory knobs of the reference object: accepted, validated as the reference validates them, and without effect --------
    def set_attention_slice(self, slice_size="auto"):
        """``unet.py:209-272``.  Sliced attention trades speed for the memory of the ``[heads, N, 2N]`` score tensor; the fused
        attention kernels never materialise that tensor, so there is nothing to slice.  The argument checks (one entry per
        attention layer -- attn1, attn2, attn_temp of the 16 transformer blocks -- each at most the head count) are the
        reference's, with its messages."""
        n_blocks = sum(1 for t in self.ucfg.down_block_types if t.startswith("CrossAttn")) * self.ucfg.layers_per_block + 1 + \
            sum(1 for t in self.ucfg.up_block_types if t.startswith("CrossAttn")) * (self.ucfg.layers_per_block + 1)
        hd = self.ucfg.attention_head_dim
        if isinstance(hd, int):
            dims = [hd] * (3 * n_blocks)
        else:                # module order of the reference: down blocks, mid block, up blocks (reversed head list), 3 attentions per block
            L = self.ucfg.layers_per_block
            per = [hd[i] for i, t in enumerate(self.ucfg.down_block_types) if t.startswith("CrossAttn") for _ in range(L)] + [hd[-1]] + \
                  [list(reversed(hd))[i] for i, t in enumerate(self.ucfg.up_block_types) if t.startswith("CrossAttn") for _ in range(L + 1)]
            dims = [h for h in per for _ in range(3)]
        if slice_size == "auto":
            slice_size = [d // 2 for d in dims]
        elif slice_size == "max":
            slice_size = len(dims) * [1]
        slice_size = len(dims) * [slice_size] if not isinstance(slice_size, list) else slice_size
        if len(slice_size) != len(dims):
            raise ValueError(
                f"You have provided {len(slice_size)}, but {self.config} has {len(dims)} different"
                f" attention layers. Make sure to match `len(slice_size)` to be {len(dims)}."
            )
        for size, dim in zip(slice_size, dims):
            if size is not None and size > dim:
                raise ValueError(f"size {size} has to be smaller or equal to {dim}.")
        self._attention_slice = list(slice_size)

    def enable_gradient_checkpointing(self):
        """``ModelMixin.enable_gradient_checkpointing`` -> ``_set_gradient_checkpointing`` (``unet.py:274-276``): a training-time
        memory knob; this object only runs inference."""
        self.gradient_checkpointing = True

    def disable_gradient_checkpointing(self):
        self.gradient_checkpointing = False

    def enable_xformers_memory_efficient_attention(self, *a, **k):      # train_finetune_videodiffusion.py:130-134
        return None

    def state_dict_spec(self):
        return unet_param_spec(self.ucfg)

    # -- weights -----------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True):
        spec = self.state_dict_spec()
        if self.use_linear_projection:           # nn.Linear proj_in / proj_out -> the 1x1-conv layout the engine's key scheme holds
            state_dict = dict(state_dict)
            for k, v in list(state_dict.items()):
                if k.endswith(".proj_in.weight") or k.endswith(".proj_out.weight"):
                    if strict and getattr(v, "ndim", 0) != 2:       # the reference's strict load refuses a conv-shaped weight for an nn.Linear
                        raise RuntimeError(f"Error(s) in loading state_dict for UNet3DConditionModel: size mismatch for {k}: copying a param "
                                           f"with shape {tuple(v.shape)} from checkpoint, the shape in current model is {tuple(v.shape[:2])}.")
                    if getattr(v, "ndim", 0) == 2:
                        state_dict[k] = v.reshape(v.shape[0], v.shape[1], 1, 1)
        missing = [k for k in spec if k not in state_dict]
        unexpected = [k for k in state_dict if k not in spec]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for UNet3DConditionModel: missing {missing[:4]}... "
                               f"unexpected {unexpected[:4]}...")
        self.engine.load_state_dict({k: v for k, v in state_dict.items() if k in spec})
        self.engine.finalize(Engine.UNET)
        return self

    def init_synthetic(self, seed: int = 42, mode: str = "reference_init"):
        """Random-init weights of this architecture from the counter RNG (no checkpoints offline)."""
        return self.load_state_dict(synth_state_dict(self.state_dict_spec(), seed=seed, mode=mode))

    @classmethod
    def _from_dir(cls, path: str, subfolder: Optional[str], inflate_2d: bool, **kw):
        if subfolder is not None:
            path = os.path.join(path, subfolder)
        config_file = os.path.join(path, "config.json")
        if not os.path.isfile(config_file):
            raise RuntimeError(f"{config_file} does not exist")                                   # unet.py:421-422
        with open(config_file) as f:
            config = json.load(f)
        config["down_block_types"] = list(UNetConfig.down_block_types)                            # unet.py:426-437
        config["up_block_types"] = list(UNetConfig.up_block_types)
        import inspect
        names = set(inspect.signature(cls.__init__).parameters) - {"self"}
        model = cls(**{k: v for k, v in config.items() if k in names}, **kw)
        model_file = os.path.join(path, WEIGHTS_NAME)
        if not os.path.isfile(model_file) and not os.path.isfile(os.path.join(path, SAFETENSORS_NAME)):
            raise RuntimeError(f"{model_file} does not exist")                                    # unet.py:442-443
        sd = _load_checkpoint(path)
        if inflate_2d:                                                                            # unet.py:445-447
            spec = model.state_dict_spec()
            fresh = synth_state_dict({k: s for k, s in spec.items() if "_temp." in k}, mode="reference_init")
            for k, v in fresh.items():
                sd[k] = torch.from_numpy(v)
        model.load_state_dict(sd)
        return model

    @classmethod
    def from_pretrained(cls, pretrained_model_path, subfolder=None, torch_dtype=None, **kw):
        """Local directory only (``inference_eeg2video.py:69``).  Checkpoints of any float type are widened to fp32 at load (the
        engine derives its own 16-bit layouts); ``torch_dtype`` selects the arithmetic as it does for the reference module:
        ``torch.float16`` -> the fp16 mode (what the reference's inference script runs), ``torch.bfloat16`` -> bf16, else fp32."""
        model = cls._from_dir(pretrained_model_path, subfolder, inflate_2d=False, **kw)
        return model.to(torch_dtype) if torch_dtype is not None else model

    @classmethod
    def from_pretrained_2d(cls, pretrained_model_path, subfolder=None, **kw):
        return cls._from_dir(pretrained_model_path, subfolder, inflate_2d=True, **kw)

    # -- forward -----------------------------------------------------------------------------
    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, class_labels=None,
                attention_mask=None, return_dict: bool = True):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is not used on the generation path")
        if torch.is_tensor(timestep):                                                            # unet.py:324-337
            tt = timestep.detach().reshape(-1).to("cpu")
            ts = tt.numpy() if tt.is_floating_point() else tt.to(torch.int64).numpy()
        else:
            ts = np.asarray([timestep])
            if not np.issubdtype(ts.dtype, np.floating):
                ts = ts.astype(np.int64)
        if ts.size not in (1, sample.shape[0]):
            raise ValueError(f"timestep has {ts.size} entries for a batch of {sample.shape[0]}")
        out = self.engine.unet_forward(sample, ts, encoder_hidden_states)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)

    __call__ = forward
