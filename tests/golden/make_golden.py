#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

Two tiers:

T1  ``EEG2Video/models/resnet.py`` needs only torch + einops and is imported directly:
    InflatedConv3d, ResnetBlock3D (Cin == Cout and Cin != Cout, with temb), Downsample3D,
    Upsample3D (scale-factor path and explicit ``output_size``; 5 -> 9 rows).
T2  ``attention.py`` / ``unet_blocks.py`` / ``unet.py`` import the pinned third-party
    package ``diffusers==0.11.1`` which is absent here (and cannot be installed: no network).
    They are executed UNMODIFIED with a test-only stand-in package that supplies the handful
    of diffusers symbols they use, written from the 0.11.1 published behaviour.
    This pins everything the reference OWNS on the path (frame gather of the sparse-causal
    attention, the (b f)/(b d) rearranges, skip bookkeeping, explicit-size upsampling, eps
    values, op order, state-dict key names).  It does NOT pin the stand-in's own primitives
    (linear + baddbmm + softmax attention, GEGLU, sinusoid): those stay "parity unpinned".

The fixtures hold inputs and expected outputs only (small .npz files); weights are
regenerated from ``eeg2video_amd.weights`` (counter RNG) by whoever consumes them.
The stand-in package is written to a temporary directory and never shipped.
"""
from __future__ import annotations

import os
import sys
import tempfile
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True          # the reference tree is read-only

import torch  # noqa: E402

from eeg2video_amd.weights import (TINY_UNET, counter_normal, synth_state_dict,  # noqa: E402
                                   synth_tensor, unet_param_spec)

_SHIM = {
    "diffusers/__init__.py": "",
    "diffusers/configuration_utils.py": '''
        import functools, inspect
        class FrozenDict(dict):
            def __getattr__(self, k):
                try: return self[k]
                except KeyError: raise AttributeError(k)
        class ConfigMixin:
            @property
            def config(self): return self._internal_dict
            @classmethod
            def from_config(cls, config, **kw):
                sig = inspect.signature(cls.__init__).parameters
                return cls(**{k: v for k, v in dict(config).items() if k in sig})
        def register_to_config(init):
            @functools.wraps(init)
            def inner(self, *args, **kwargs):
                sig = inspect.signature(init)
                names = [n for n in sig.parameters if n != "self"]
                cfg = {n: p.default for n, p in sig.parameters.items() if n != "self"}
                cfg.update(dict(zip(names, args))); cfg.update(kwargs)
                object.__setattr__(self, "_internal_dict", FrozenDict(cfg))
                init(self, *args, **kwargs)
            return inner
    ''',
    "diffusers/modeling_utils.py": '''
        import torch
        class ModelMixin(torch.nn.Module):
            @property
            def dtype(self): return next(self.parameters()).dtype
            @property
            def device(self): return next(self.parameters()).device
    ''',
    "diffusers/utils/__init__.py": '''
        import logging as _logging
        WEIGHTS_NAME = "diffusion_pytorch_model.bin"
        class BaseOutput:
            def __getitem__(self, k):
                return getattr(self, k) if isinstance(k, str) else tuple(self.__dict__.values())[k]
        class logging:
            @staticmethod
            def get_logger(name): return _logging.getLogger(name)
    ''',
    "diffusers/utils/import_utils.py": "def is_xformers_available():\n    return False\n",
    "diffusers/models/__init__.py": "",
    # diffusers 0.11.1 models/attention.py: CrossAttention / FeedForward / GEGLU
    "diffusers/models/attention.py": '''
        import torch, torch.nn as nn, torch.nn.functional as F
        class CrossAttention(nn.Module):
            def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0,
                         bias=False, upcast_attention=False, upcast_softmax=False,
                         added_kv_proj_dim=None, norm_num_groups=None):
                super().__init__()
                inner = dim_head * heads
                cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
                self.upcast_attention, self.upcast_softmax = upcast_attention, upcast_softmax
                self.scale = dim_head ** -0.5
                self.heads = heads
                self.sliceable_head_dim = heads
                self._slice_size = None
                self._use_memory_efficient_attention_xformers = False
                self.added_kv_proj_dim = added_kv_proj_dim
                self.group_norm = None
                self.to_q = nn.Linear(query_dim, inner, bias=bias)
                self.to_k = nn.Linear(cross_attention_dim, inner, bias=bias)
                self.to_v = nn.Linear(cross_attention_dim, inner, bias=bias)
                self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])
            def reshape_heads_to_batch_dim(self, t):
                b, s, d = t.shape; h = self.heads
                return t.reshape(b, s, h, d // h).permute(0, 2, 1, 3).reshape(b * h, s, d // h)
            def reshape_batch_dim_to_heads(self, t):
                b, s, d = t.shape; h = self.heads
                return t.reshape(b // h, h, s, d).permute(0, 2, 1, 3).reshape(b // h, s, d * h)
            def _attention(self, q, k, v, attention_mask=None):
                if self.upcast_attention: q, k = q.float(), k.float()
                s = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device),
                                  q, k.transpose(-1, -2), beta=0, alpha=self.scale)
                if attention_mask is not None: s = s + attention_mask
                if self.upcast_softmax: s = s.float()
                p = s.softmax(dim=-1).to(v.dtype)
                return self.reshape_batch_dim_to_heads(torch.bmm(p, v))
            def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None):
                q = self.reshape_heads_to_batch_dim(self.to_q(hidden_states))
                ctx = encoder_hidden_states if encoder_hidden_states is not None else hidden_states
                k = self.reshape_heads_to_batch_dim(self.to_k(ctx))
                v = self.reshape_heads_to_batch_dim(self.to_v(ctx))
                o = self._attention(q, k, v, attention_mask)
                return self.to_out[1](self.to_out[0](o))
        class GEGLU(nn.Module):
            def __init__(self, dim_in, dim_out):
                super().__init__(); self.proj = nn.Linear(dim_in, dim_out * 2)
            def forward(self, x):
                h, gate = self.proj(x).chunk(2, dim=-1)
                return h * F.gelu(gate)
        class FeedForward(nn.Module):
            def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu"):
                super().__init__()
                assert activation_fn == "geglu"
                inner = int(dim * mult); dim_out = dim_out if dim_out is not None else dim
                self.net = nn.ModuleList([GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim_out)])
            def forward(self, x):
                for m in self.net: x = m(x)
                return x
        class AdaLayerNorm(nn.Module):
            def __init__(self, *a, **k): raise NotImplementedError
    ''',
    # diffusers 0.11.1 models/embeddings.py
    "diffusers/models/embeddings.py": '''
        import math, torch, torch.nn as nn
        def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1,
                                   scale=1, max_period=10000):
            half = embedding_dim // 2
            exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
            exponent = exponent / (half - downscale_freq_shift)
            emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
            emb = scale * emb
            emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
            if flip_sin_to_cos: emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
            return emb
        class Timesteps(nn.Module):
            def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
                super().__init__()
                self.num_channels, self.flip, self.shift = num_channels, flip_sin_to_cos, downscale_freq_shift
            def forward(self, t):
                return get_timestep_embedding(t, self.num_channels, self.flip, self.shift)
        class TimestepEmbedding(nn.Module):
            def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None):
                super().__init__()
                self.linear_1 = nn.Linear(in_channels, time_embed_dim); self.act = nn.SiLU()
                self.linear_2 = nn.Linear(time_embed_dim, out_dim or time_embed_dim)
            def forward(self, x): return self.linear_2(self.act(self.linear_1(x)))
    ''',
}


def _write_shim(root: str) -> None:
    for rel, body in _SHIM.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(textwrap.dedent(body))


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _load(module: torch.nn.Module, prefix: str, seed: int):
    """Fill ``module`` from the counter RNG using ``prefix + <param name>`` as the stream key."""
    spec = {prefix + k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth_state_dict(spec, seed=seed, mode="perturbed")
    module.load_state_dict({k[len(prefix):]: _t(v) for k, v in sd.items()}, strict=True)
    return sd


def tier1(out: dict) -> None:
    sys.path.insert(0, os.path.join(REF, "EEG2Video", "models"))
    import resnet as ref_resnet            # the reference file itself

    torch.manual_seed(0)
    f, h, w = 3, 5, 8
    x32 = _t(counter_normal(7, "t1.x32", (2, 32, f, h, w)))
    x64 = _t(counter_normal(7, "t1.x64", (2, 64, f, h, w)))
    temb = _t(counter_normal(7, "t1.temb", (2, 128)))
    out["t1.x32"], out["t1.x64"], out["t1.temb"] = x32.numpy(), x64.numpy(), temb.numpy()

    conv = ref_resnet.InflatedConv3d(32, 64, 3, padding=1)
    _load(conv, "t1.conv.", 11)
    out["t1.conv.out"] = conv(x32).detach().numpy()

    for tag, cin, cout, x in (("same", 64, 64, x64), ("proj", 32, 64, x32)):
        blk = ref_resnet.ResnetBlock3D(in_channels=cin, out_channels=cout, temb_channels=128, groups=8, eps=1e-5)
        _load(blk, f"t1.res_{tag}.", 11)
        out[f"t1.res_{tag}.out"] = blk(x, temb).detach().numpy()

    down = ref_resnet.Downsample3D(64, use_conv=True, out_channels=64, padding=1, name="op")
    sd = {k: v for k, v in down.state_dict().items()}
    # `conv` is the only module for name="op" (resnet.py:96-97)
    _load(down, "t1.down.", 11)
    out["t1.down.out"] = down(x64).detach().numpy()

    up = ref_resnet.Upsample3D(64, use_conv=True, out_channels=64)
    _load(up, "t1.up.", 11)
    out["t1.up.out_x2"] = up(x64).detach().numpy()                        # (f,5,8) -> (f,10,16)
    out["t1.up.out_9x16"] = up(x64, output_size=(f, 9, 16)).detach().numpy()   # rows 0,0,1,1,2,2,3,3,4
    out["t1.up.out_7x11"] = up(x64, output_size=(f, 7, 11)).detach().numpy()   # non-uniform both ways


def tier2(out: dict) -> None:
    shim_root = tempfile.mkdtemp(prefix="e2v_dep_standin_")
    _write_shim(shim_root)
    sys.path.insert(0, shim_root)
    sys.path.insert(0, REF)
    from EEG2Video.models.unet import UNet3DConditionModel                 # reference, unmodified
    from EEG2Video.models.attention import Transformer3DModel

    cfg = TINY_UNET
    model = UNet3DConditionModel(
        sample_size=cfg.sample_size, in_channels=4, out_channels=4, block_out_channels=cfg.block_out_channels,
        layers_per_block=cfg.layers_per_block, cross_attention_dim=cfg.cross_attention_dim,
        attention_head_dim=cfg.attention_head_dim, norm_num_groups=cfg.norm_num_groups, norm_eps=cfg.norm_eps)
    model.eval()
    spec = unet_param_spec(cfg)
    ref_keys = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    assert ref_keys == dict(spec), "state-dict key scheme differs from the reference's"
    sd = synth_state_dict(spec, seed=42, mode="perturbed")
    model.load_state_dict({k: _t(v) for k, v in sd.items()}, strict=True)

    # explicit-size upsample path: 9 -> 5 -> 3 -> 2 rows, 12 -> 6 -> 3 -> 2 cols
    x = _t(counter_normal(1234, "t2.latent", (2, 4, 3, 9, 12)))
    cond = _t(counter_normal(1235, "t2.cond", (2, 11, cfg.cross_attention_dim)))
    taps = {}

    def hook(name):
        def fn(_m, _i, o):
            taps[name] = (o[0] if isinstance(o, tuple) else o)
        return fn
    for i, blk in enumerate(model.down_blocks):
        blk.register_forward_hook(hook(f"down{i}"))
    model.mid_block.register_forward_hook(hook("mid"))
    for i, blk in enumerate(model.up_blocks):
        blk.register_forward_hook(hook(f"up{i}"))
    with torch.no_grad():
        y = model(x, 501, encoder_hidden_states=cond).sample
        taps = {k: v.detach().clone() for k, v in taps.items()}      # the second call re-fires the hooks
        hooks_done = dict(taps)
        y_vec = model(x, torch.tensor([751, 1]), encoder_hidden_states=cond)["sample"]
    out["t2.unet.x"], out["t2.unet.cond"] = x.numpy(), cond.numpy()
    out["t2.unet.out_t501"] = y.numpy()
    out["t2.unet.out_t751_1"] = y_vec.numpy()
    for k, v in hooks_done.items():
        out[f"t2.unet.tap.{k}"] = v.numpy()

    # one Transformer3DModel on its own, 8 heads x 8 dims, 4 frames (sparse-causal gather with f >= 3)
    tr = Transformer3DModel(8, 8, in_channels=64, num_layers=1, cross_attention_dim=cfg.cross_attention_dim,
                            norm_num_groups=32)
    tr.eval()
    _load(tr, "t2.tr.", 13)
    xt = _t(counter_normal(99, "t2.tr.x", (2, 64, 4, 5, 6)))
    ct = _t(counter_normal(99, "t2.tr.cond", (2, 7, cfg.cross_attention_dim)))
    with torch.no_grad():
        out["t2.tr.out"] = tr(xt, encoder_hidden_states=ct).sample.numpy()
    out["t2.tr.x"], out["t2.tr.cond"] = xt.numpy(), ct.numpy()


def tier2_linear_projection(out: dict) -> None:
    """Transformer3DModel(use_linear_projection=True) (attention.py:60-63,83-86,99-123: proj_in / proj_out as nn.Linear on the tokens,
    the SD-2.x form) run unmodified: pins that the mirror's handling of the option -- the two [C, C] weights reshaped to the 1x1-conv
    layout, same GEMM on channel-last rows -- is what the reference computes."""
    shim_root = tempfile.mkdtemp(prefix="e2v_dep_standin_")
    _write_shim(shim_root)
    sys.path.insert(0, shim_root)
    sys.path.insert(0, REF)
    from EEG2Video.models.attention import Transformer3DModel
    cfg = TINY_UNET
    tr = Transformer3DModel(8, 8, in_channels=64, num_layers=1, cross_attention_dim=cfg.cross_attention_dim,
                            norm_num_groups=32, use_linear_projection=True)
    tr.eval()
    sd = _load(tr, "t2.trlin.", 17)
    assert sd["t2.trlin.proj_in.weight"].ndim == 2 and sd["t2.trlin.proj_out.weight"].ndim == 2
    xt = _t(counter_normal(98, "t2.trlin.x", (2, 64, 3, 5, 6)))
    ct = _t(counter_normal(98, "t2.trlin.cond", (2, 7, cfg.cross_attention_dim)))
    with torch.no_grad():
        out["t2.trlin.out"] = tr(xt, encoder_hidden_states=ct).sample.numpy()
    out["t2.trlin.x"], out["t2.trlin.cond"] = xt.numpy(), ct.numpy()
    for k, v in sd.items():                     # every weight travels with the fixture (the Linear-shaped ones have no spec here)
        out["t2.trlin.w." + k[len("t2.trlin."):]] = v


def tier1_extras(out: dict) -> None:
    """`CLIP` (semantic predictor) and DANA `Diffusion`, both import with torch alone."""
    from eeg2video_amd.weights import SemanticConfig, semantic_param_spec
    sys.path.insert(0, os.path.join(REF, "EEG2Video", "models"))
    import train_semantic_predictor as ref_sem          # the reference file itself (no code runs at import)
    import DANA_module as ref_dana

    model = ref_sem.CLIP().eval()                       # 310 -> 10000 x4 -> 59136, 0.89 G parameters
    spec = semantic_param_spec(SemanticConfig(), 768)
    assert {k: tuple(v.shape) for k, v in model.state_dict().items()} == dict(spec), "CLIP key scheme differs"
    sd = synth_state_dict(spec, seed=44, mode="reference_init")
    model.load_state_dict({k: _t(v) for k, v in sd.items()}, strict=True)
    eeg = _t(counter_normal(310, "t1.clip.eeg", (2, 310)))
    with torch.no_grad():
        y = model(eeg).numpy()
    idx = (np.arange(2048, dtype=np.int64) * 7919) % y.shape[1]
    out["x.clip.eeg"], out["x.clip.idx"] = eeg.numpy(), idx
    out["x.clip.out_sampled"] = y[:, idx]
    out["x.clip.out_l2"] = np.sqrt((y.astype(np.float64) ** 2).sum(axis=1))
    del model, sd

    diff = ref_dana.Diffusion(time_steps=500)
    out["x.dana.sqrt_alphas_cumprod"] = diff.sqrt_alphas_cumprod.numpy()
    out["x.dana.sqrt_one_minus_alphas_cumprod"] = diff.sqrt_one_minus_alphas_cumprod.numpy()
    # Diffusion.forward draws t / noise itself and calls .cuda(); run it unmodified with the draws pinned
    b, f, c, h, w = 3, 6, 4, 5, 8
    x0 = _t(counter_normal(1, "x.dana.x0", (b, f, c, h, w)))
    ed = _t(counter_normal(2, "x.dana.ed", (b, f, c, h, w)))
    es = _t(counter_normal(3, "x.dana.es", (b, 1, c, h, w)))
    t = torch.tensor([0, 137, 499])
    saved = (torch.randint, torch.randn_like, torch.randn, torch.Tensor.cuda)
    try:
        torch.randint = lambda *a, **k: t.clone()
        torch.randn_like = lambda *a, **k: ed.clone()
        torch.randn = lambda *a, **k: es.clone()
        torch.Tensor.cuda = lambda self, *a, **k: self
        diff.device = torch.device("cpu")
        y = diff.forward(x0, 0.3)
    finally:
        torch.randint, torch.randn_like, torch.randn, torch.Tensor.cuda = saved
    out["x.dana.x0"], out["x.dana.eps_div"], out["x.dana.eps_same"] = x0.numpy(), ed.numpy(), es.numpy()
    out["x.dana.t"], out["x.dana.out_beta0.3"] = t.numpy(), y.numpy()


def tier1_inversion(out: dict) -> None:
    """DDIM inversion, `EEG2Video_New/Generation/tuneavideo/util.py:56-101`.  The file imports
    `imageio` and `torchvision` (absent here) for its GIF writer only; empty stand-in modules let it import, and
    `next_step` / `ddim_loop` then run unmodified.  `ddim_loop` reads `cond_embeddings.pt` and moves it to 'cuda' in
    fp16 (:80-82): `torch.load` is pinned to our tensor and the move is redirected to the CPU (dtype kept)."""
    import types
    for name in ("imageio", "torchvision"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, os.path.join(REF, "EEG2Video_New", "Generation", "tuneavideo"))
    import util as ref_util

    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2      # SD-v1-4 scheduler config
    abar = torch.cumprod(1.0 - betas, dim=0)
    out["inv.alphas_cumprod"] = abar.numpy()

    def sched(n):
        ts = (np.arange(0, n) * (1000 // n)).round()[::-1].copy().astype(np.int64) + 1
        return types.SimpleNamespace(config=types.SimpleNamespace(num_train_timesteps=1000), num_inference_steps=n,
                                     alphas_cumprod=abar, final_alpha_cumprod=abar[0], timesteps=torch.from_numpy(ts))

    eps = _t(counter_normal(21, "inv.eps", (2, 4, 3, 5, 6)))
    x = _t(counter_normal(22, "inv.x", (2, 4, 3, 5, 6)))
    out["inv.eps"], out["inv.x"] = eps.numpy(), x.numpy()
    cases = []
    for n in (50, 4, 333, 20):
        s = sched(n)
        for t in sorted({int(s.timesteps[-1]), int(s.timesteps[len(s.timesteps) // 2]), int(s.timesteps[0])}):
            cases.append((n, t))
            out[f"inv.next_step.n{n}.t{t}"] = ref_util.next_step(eps, t, x, s).numpy()
    out["inv.next_step.cases"] = np.array(cases, dtype=np.int64)

    # the loop, with a closed-form stand-in for the UNet (the real one is pinned by tier 2)
    cond = _t(counter_normal(23, "inv.cond", (1, 7, 16)))

    def unet(latents, t, encoder_hidden_states):
        c = encoder_hidden_states.float().mean(dim=(1, 2)).view(-1, 1, 1, 1, 1)
        return {"sample": 0.3 * latents + 0.05 * torch.sin(latents * 3.0) + c + float(t) * 1e-4}

    n = 5
    s = sched(n)
    saved = (torch.load, torch.Tensor.to)

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if (isinstance(v, str) and v.startswith("cuda")) else v for v in a)
        return saved[1](self, *a, **k)

    try:
        torch.load = lambda *a, **k: cond.clone()
        torch.Tensor.to = to_cpu
        lat = ref_util.ddim_inversion(unet, s, x, n, prompt="")
    finally:
        torch.load, torch.Tensor.to = saved
    out["inv.loop.cond"], out["inv.loop.n"] = cond.numpy(), np.array([n], dtype=np.int64)
    out["inv.loop.latents"] = np.stack([l.numpy() for l in lat])


def main() -> None:
    if "--only-inversion" in sys.argv:                      # add one fixture without re-running the 0.9 G-parameter CLIP
        ti = {}
        tier1_inversion(ti)
        np.savez_compressed(os.path.join(HERE, "reference_t1_inversion.npz"), **ti)
        print("reference_t1_inversion.npz", os.path.getsize(os.path.join(HERE, "reference_t1_inversion.npz")) // 1024, "KiB")
        return
    if "--only-linear-projection" in sys.argv:              # add this fixture alone (the others stay byte for byte what they were)
        tl = {}
        tier2_linear_projection(tl)
        np.savez_compressed(os.path.join(HERE, "reference_t2_linear_projection.npz"), **tl)
        print("reference_t2_linear_projection.npz", os.path.getsize(os.path.join(HERE, "reference_t2_linear_projection.npz")) // 1024, "KiB")
        return
    t1, t2, tx = {}, {}, {}
    tier2_linear_projection(tl := {})
    np.savez_compressed(os.path.join(HERE, "reference_t2_linear_projection.npz"), **tl)
    tier1_inversion(t1i := {})
    np.savez_compressed(os.path.join(HERE, "reference_t1_inversion.npz"), **t1i)
    tier1(t1)
    tier2(t2)
    tier1_extras(tx)
    np.savez_compressed(os.path.join(HERE, "reference_t1_extras.npz"), **tx)
    print("reference_t1_extras.npz", os.path.getsize(os.path.join(HERE, "reference_t1_extras.npz")) // 1024, "KiB")
    np.savez_compressed(os.path.join(HERE, "reference_t1_resnet.npz"), **t1)
    np.savez_compressed(os.path.join(HERE, "reference_t2_unet_tiny.npz"), **t2)
    for name in ("reference_t1_resnet.npz", "reference_t2_unet_tiny.npz"):
        print(name, os.path.getsize(os.path.join(HERE, name)) // 1024, "KiB")


if __name__ == "__main__":
    main()
