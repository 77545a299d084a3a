"""``Engine``: one ``e2v_ctx`` (weights + workspace on one GPU) behind a small Python object.

torch is used for what the boundary needs and nothing else: device buffers (``torch.empty``),
the current HIP stream, and D2H copies.  Every computation is a call into ``libeeg2video_hip.so``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Iterable, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .weights import SemanticConfig, UNetConfig, VAEConfig

_ERRORS = {
    _lib.E2V_EINVAL: ValueError,
    _lib.E2V_ESHAPE: ValueError,
    _lib.E2V_ENOWEIGHT: RuntimeError,
    _lib.E2V_EHIP: RuntimeError,
    _lib.E2V_ESTATE: RuntimeError,
}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class Engine:
    """Owns one ``e2v_ctx``.  ``unet_cfg`` / ``vae_cfg`` mirror the reference configs."""

    UNET, VAE, SEMANTIC = 1, 2, 4

    def __init__(self, unet_cfg: UNetConfig = UNetConfig(), vae_cfg: VAEConfig = VAEConfig(), device: int = 0,
                 sem_cfg: SemanticConfig = SemanticConfig()):
        if not torch.cuda.is_available():
            raise RuntimeError("eeg2video_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                               "there is no CPU path")
        self.lib = _lib.load()
        self.unet_cfg, self.vae_cfg, self.sem_cfg = unet_cfg, vae_cfg, sem_cfg
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        torch.zeros(1, device=self.device)          # make sure torch has initialised the device's context
        cfg = _lib.E2VConfig()
        self.lib.e2v_default_config(C.byref(cfg))
        cfg.in_channels, cfg.out_channels = unet_cfg.in_channels, unet_cfg.out_channels
        cfg.block_out_channels = (C.c_int * 4)(*unet_cfg.block_out_channels)
        cfg.layers_per_block = unet_cfg.layers_per_block
        cfg.cross_attention_dim = unet_cfg.cross_attention_dim
        hd = unet_cfg.attention_head_dim                      # an int, or one head count per down block (unet.py:71,110-111)
        if isinstance(hd, int):
            cfg.attention_heads = hd
        else:
            if len(hd) != 4:
                raise ValueError("attention_head_dim as a tuple needs one entry per down block (4)")
            cfg.attention_heads = int(hd[0])
            cfg.attention_heads_per_block = (C.c_int * 4)(*[int(h) for h in hd])
        cfg.norm_num_groups, cfg.norm_eps = unet_cfg.norm_num_groups, unet_cfg.norm_eps
        cfg.flip_sin_to_cos, cfg.freq_shift = int(unet_cfg.flip_sin_to_cos), float(unet_cfg.freq_shift)
        cfg.vae_in_channels, cfg.vae_latent_channels = vae_cfg.in_channels, vae_cfg.latent_channels
        cfg.vae_block_out_channels = (C.c_int * 4)(*vae_cfg.block_out_channels)
        cfg.vae_layers_per_block = vae_cfg.layers_per_block
        cfg.vae_norm_num_groups, cfg.vae_norm_eps = vae_cfg.norm_num_groups, vae_cfg.norm_eps
        cfg.vae_scaling_factor = vae_cfg.scaling_factor
        cfg.sem_in_features, cfg.sem_hidden, cfg.sem_tokens = sem_cfg.in_features, sem_cfg.hidden, sem_cfg.tokens
        self._cfg = cfg
        ctx = C.c_void_p()
        st = self.lib.e2v_create(C.byref(cfg), device, C.byref(ctx))
        if st != _lib.E2V_OK:
            raise _ERRORS.get(st, RuntimeError)(self.lib.e2v_last_error(None).decode())
        self.ctx = ctx
        self.ready = 0
        self.compute_dtype = "fp32"

    def __del__(self):
        ctx, self.ctx = getattr(self, "ctx", None), None
        if ctx:
            self.lib.e2v_destroy(ctx)

    # ------------------------------------------------------------------ helpers
    def _check(self, st: int) -> None:
        if st != _lib.E2V_OK:
            raise _ERRORS.get(st, RuntimeError)(self.lib.e2v_last_error(self.ctx).decode())

    def _dev(self, t: torch.Tensor, name: str) -> torch.Tensor:
        if not isinstance(t, torch.Tensor):
            raise ValueError(f"`{name}` has to be of type `torch.Tensor` but is {type(t)}")
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    def expected_keys(self) -> Dict[str, Tuple[int, ...]]:
        out = {}
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(self.lib.e2v_num_expected_keys(self.ctx)):
            k = self.lib.e2v_expected_key(self.ctx, i, shape, C.byref(nd)).decode()
            out[k] = tuple(int(shape[d]) for d in range(nd.value))
        return out

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd: Mapping[str, object], prefix: str = "") -> None:
        """Upload tensors keyed by the reference's state-dict names (``prefix='vae.'`` for the VAE)."""
        for k, v in sd.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu()
                if v.dtype not in (torch.float32, torch.float16):      # bf16 / fp64 checkpoints: widened (or narrowed) to fp32 here
                    v = v.float()
                v = v.numpy()
            a = np.ascontiguousarray(v)
            if a.dtype == np.float16:
                dt = _lib.E2V_F16
            else:
                a = np.ascontiguousarray(a, dtype=np.float32)
                dt = _lib.E2V_F32
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(self.lib.e2v_load_tensor(self.ctx, (prefix + k).encode(), a.ctypes.data_as(C.c_void_p), dt,
                                                 shape, a.ndim))

    def finalize(self, which: int) -> None:
        self._check(self.lib.e2v_finalize_weights(self.ctx, which))
        self.ready |= which

    def set_compute_dtype(self, dtype) -> None:
        """'fp32' (default, parity configuration), 'bf16' (BASELINE configs[2]: bf16 MFMA, bf16 activations in HBM, fp32 accumulate /
        statistics / softmax) or 'fp16' (the same kernels on IEEE half: the reference's own inference dtype,
        ``inference_eeg2video.py:69-70`` ``torch_dtype=torch.float16``).  A ``torch.dtype`` is accepted too."""
        code = compute_dtype_code(dtype)
        self._check(self.lib.e2v_set_compute_dtype(self.ctx, code))
        self.compute_dtype = {_lib.E2V_F32: "fp32", _lib.E2V_F16: "fp16", _lib.E2V_BF16: "bf16", _lib.E2V_F32X3: "f32x3"}[code]

    def set_conv_algo(self, algo: str) -> None:
        """'auto' (default: Winograd F(2x2,3x3) for the wide stride-1 3x3 convs, direct implicit GEMM elsewhere),
        'direct' or 'winograd'.  For the graph entry points call it before the weights are finalized."""
        code = {"auto": 0, "direct": 1, "winograd": 2, "winograd4": 3}[algo]
        self._check(self.lib.e2v_set_conv_algo(self.ctx, code))

    def set_knob(self, name: str, value: int) -> None:
        """Run-time switch of DESIGN section 10 (tests / profiling: same-process A/B of kernel variants)."""
        if self.lib.e2v_op_set_knob(name.encode(), int(value)) != 0:
            raise ValueError(f"unknown switch {name!r}")

    def device_bytes(self) -> int:
        return int(self.lib.e2v_device_bytes(self.ctx))

    def profile_begin(self) -> None:
        self._check(self.lib.e2v_profile_begin(self.ctx))

    def profile_end(self) -> dict:
        import json
        buf = C.create_string_buffer(1 << 20)
        n = self.lib.e2v_profile_end(self.ctx, buf, len(buf))
        if n < 0:
            raise RuntimeError("e2v_profile_end failed")
        return json.loads(buf.value.decode())

    # ------------------------------------------------------------------ schedule
    def ddim_timesteps(self, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.int64)
        self._check(self.lib.e2v_ddim_timesteps(self.ctx, n, out.ctypes.data_as(_lib.c_int64_p)))
        return out

    def alphas_cumprod(self) -> np.ndarray:
        out = np.empty(self._cfg.num_train_timesteps, dtype=np.float32)
        self._check(self.lib.e2v_ddim_alphas_cumprod(self.ctx, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def set_alphas_cumprod(self, table: np.ndarray) -> None:
        t = np.ascontiguousarray(table, dtype=np.float32)
        self._check(self.lib.e2v_set_alphas_cumprod(self.ctx, t.ctypes.data_as(C.POINTER(C.c_float)), t.size))

    def set_ddim_schedule(self, alphas_cumprod: np.ndarray, steps_offset: int) -> None:
        """Hand the ctx the schedule of the scheduler object the caller holds (table + ``steps_offset``), so that the
        fused loop and a stepped loop driven by that scheduler agree."""
        t = np.ascontiguousarray(alphas_cumprod, dtype=np.float32)
        self._check(self.lib.e2v_set_ddim_schedule(self.ctx, t.ctypes.data_as(C.POINTER(C.c_float)), t.size, int(steps_offset)))
        self._cfg.num_train_timesteps, self._cfg.steps_offset = int(t.size), int(steps_offset)

    def _check_loop_inputs(self, x: torch.Tensor, cond: torch.Tensor, un: Optional[torch.Tensor]) -> None:
        """Shapes the fused loops index device memory by: everything is checked here, on the host, because the library
        copies ``B*T*D`` floats from ``cond`` / ``uncond`` whatever their real size."""
        if x.dim() != 5 or cond.dim() != 3:
            raise ValueError(f"expected latents [B,C,F,h,w] and cond [B,T,D], got {tuple(x.shape)} / {tuple(cond.shape)}")
        b, c = x.shape[0], x.shape[1]
        if c != self.unet_cfg.in_channels:
            raise ValueError(f"latents have {c} channels, the UNet takes {self.unet_cfg.in_channels}")
        if cond.shape[0] != b:
            raise ValueError(f"cond batch {cond.shape[0]} != latent batch {b}")
        if cond.shape[2] != self.unet_cfg.cross_attention_dim:
            raise ValueError(f"cond feature dim {cond.shape[2]} != cross_attention_dim {self.unet_cfg.cross_attention_dim}")
        if un is not None:
            if un.dim() != 3 or un.shape[0] not in (1, b) or tuple(un.shape[1:]) != tuple(cond.shape[1:]):
                raise ValueError(f"uncond must be [1 or {b},{cond.shape[1]},{cond.shape[2]}], got {tuple(un.shape)}")

    # ------------------------------------------------------------------ hot path
    def unet_forward(self, sample: torch.Tensor, timesteps: Sequence[int], cond: torch.Tensor) -> torch.Tensor:
        sample, cond = self._dev(sample, "sample"), self._dev(cond, "encoder_hidden_states")
        if sample.dim() != 5 or cond.dim() != 3 or cond.shape[0] != sample.shape[0]:
            raise ValueError(f"expected sample [N,C,F,H,W] and cond [N,T,D], got {tuple(sample.shape)} / {tuple(cond.shape)}")
        n, c, f, h, w = sample.shape
        if c != self.unet_cfg.in_channels or cond.shape[2] != self.unet_cfg.cross_attention_dim:
            raise ValueError("channel count of `sample` or feature dim of `encoder_hidden_states` does not match the config")
        out = torch.empty((n, self.unet_cfg.out_channels, f, h, w), device=self.device, dtype=torch.float32)
        tarr = np.asarray(timesteps).reshape(-1)
        if np.issubdtype(tarr.dtype, np.floating) and not np.all(tarr == np.round(tarr)):
            tf = np.ascontiguousarray(tarr, dtype=np.float32)        # fractional timesteps (Euler / LMS schedules)
            self._check(self.lib.e2v_unet_forward_ft(self.ctx, sample.data_ptr(), tf.ctypes.data_as(C.POINTER(C.c_float)), tf.size,
                                                     cond.data_ptr(), n, f, h, w, cond.shape[1], out.data_ptr(), _stream()))
            return out
        ts = np.ascontiguousarray(tarr.astype(np.int64))
        self._check(self.lib.e2v_unet_forward(self.ctx, sample.data_ptr(), ts.ctypes.data_as(_lib.c_int64_p), ts.size,
                                              cond.data_ptr(), n, f, h, w, cond.shape[1], out.data_ptr(), _stream()))
        return out

    TAP_NAMES = ("emb", "down0", "down1", "down2", "down3", "mid", "up0", "up1", "up2", "up3")

    def unet_forward_taps(self, sample: torch.Tensor, timesteps: Sequence[int], cond: torch.Tensor):
        """Test aid (``e2v_op_unet_forward_taps``): the forward plus the block outputs ``oracle/unet3d.py`` exposes as ``taps``
        (fp32 NCFHW; ``emb`` as [N, 1280]).  Returns ``(sample_out, {name: tensor})``."""
        sample, cond = self._dev(sample, "sample"), self._dev(cond, "encoder_hidden_states")
        n, c, f, h, w = sample.shape
        out = torch.empty((n, self.unet_cfg.out_channels, f, h, w), device=self.device, dtype=torch.float32)
        boc = list(self.unet_cfg.block_out_channels)
        down = lambda v: (v - 1) // 2 + 1                              # 3x3, stride 2, padding 1
        hs, ws = [h], [w]
        for _ in range(3):
            hs.append(down(hs[-1])); ws.append(down(ws[-1]))
        lv = lambda ch, l: n * ch * f * hs[l] * ws[l]
        cap = (n * 4 * boc[0] + sum(lv(boc[i], min(i + 1, 3)) for i in range(4)) + lv(boc[3], 3)
               + sum(lv(boc[3 - i], max(3 - i - 1, 0)) for i in range(4)))
        buf = torch.empty(int(cap), device=self.device, dtype=torch.float32)
        shapes = np.zeros(80, dtype=np.int64)
        count = C.c_int(0)
        ts = np.ascontiguousarray(np.asarray(timesteps).reshape(-1).astype(np.int64))
        self._check(self.lib.e2v_op_unet_forward_taps(self.ctx, sample.data_ptr(), ts.ctypes.data_as(_lib.c_int64_p), ts.size,
                                                      cond.data_ptr(), n, f, h, w, cond.shape[1], out.data_ptr(), buf.data_ptr(),
                                                      buf.numel(), shapes.ctypes.data_as(_lib.c_int64_p), C.byref(count), _stream()))
        taps, off = {}, 0
        for i in range(count.value):
            shp = [int(v) for v in shapes[5 * i:5 * i + 5]]
            cnt = int(np.prod(shp))
            t = buf[off:off + cnt].reshape(shp)
            taps[self.TAP_NAMES[i]] = t.reshape(shp[0], shp[1]) if self.TAP_NAMES[i] == "emb" else t
            off += cnt
        return out, taps

    def ddim_cfg_step(self, eps_uncond: torch.Tensor, eps_cond: Optional[torch.Tensor], x: torch.Tensor,
                      guidance_scale: float, t: int, t_prev: int) -> torch.Tensor:
        eu, x = self._dev(eps_uncond, "eps"), self._dev(x, "sample")
        ec = self._dev(eps_cond, "eps_cond") if eps_cond is not None else None
        out = torch.empty_like(x)
        self._check(self.lib.e2v_ddim_cfg_step(self.ctx, eu.data_ptr(), ec.data_ptr() if ec is not None else None,
                                               x.data_ptr(), out.data_ptr(), x.numel(), float(guidance_scale), int(t),
                                               int(t_prev), _stream()))
        return out

    def cfg_combine(self, eps_uncond: torch.Tensor, eps_cond: torch.Tensor, guidance_scale: float) -> torch.Tensor:
        eu, ec = self._dev(eps_uncond, "eps_uncond"), self._dev(eps_cond, "eps_cond")
        out = torch.empty_like(eu)
        self._check(self.lib.e2v_cfg_combine(self.ctx, eu.data_ptr(), ec.data_ptr(), float(guidance_scale), out.data_ptr(),
                                             eu.numel(), _stream()))
        return out

    def lincomb(self, terms) -> torch.Tensor:
        """sum of coef * tensor over 1..5 (coef, tensor) pairs of equal shape."""
        xs = [self._dev(t, "term") for _, t in terms]
        n = len(xs)
        if not 1 <= n <= 5 or any(x.shape != xs[0].shape for x in xs):
            raise ValueError("lincomb takes 1..5 tensors of one shape")
        out = torch.empty_like(xs[0])
        ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in xs])
        coefs = (C.c_float * n)(*[float(c) for c, _ in terms])
        self._check(self.lib.e2v_lincomb(self.ctx, n, ptrs, coefs, out.data_ptr(), out.numel(), _stream()))
        return out

    def ddim_next_step(self, eps: torch.Tensor, t: int, x: torch.Tensor, num_inference_steps: int) -> torch.Tensor:
        e, x = self._dev(eps, "eps"), self._dev(x, "sample")
        out = torch.empty_like(x)
        self._check(self.lib.e2v_ddim_next_step(self.ctx, e.data_ptr(), x.data_ptr(), out.data_ptr(), x.numel(), int(t),
                                                int(num_inference_steps), _stream()))
        return out

    def ddim_invert(self, latents: torch.Tensor, cond: torch.Tensor, num_inv_steps: int, return_all: bool = True):
        """DDIM inversion loop on the device (reference: tuneavideo/util.py ddim_loop); returns the list of n+1 latents
        (return_all) or only the last one."""
        x, cond = self._dev(latents, "latents"), self._dev(cond, "cond")
        self._check_loop_inputs(x, cond, None)
        b, c, f, h, w = x.shape
        allb = torch.empty((num_inv_steps + 1,) + tuple(x.shape), device=self.device, dtype=torch.float32) if return_all else None
        last = torch.empty_like(x) if not return_all else None
        self._check(self.lib.e2v_ddim_invert(self.ctx, x.data_ptr(), cond.data_ptr(), b, f, h, w, cond.shape[1],
                                             int(num_inv_steps), allb.data_ptr() if return_all else None,
                                             last.data_ptr() if last is not None else None, _stream()))
        return list(allb.unbind(0)) if return_all else last

    def vae_decode(self, latents: torch.Tensor, postprocess: bool = True) -> torch.Tensor:
        z = self._dev(latents, "latents")
        if z.dim() == 4:
            z = z[:, :, None]
            squeeze = True
        else:
            squeeze = False
        b, c, f, h, w = z.shape
        out = torch.empty((b, self.vae_cfg.out_channels, f, 8 * h, 8 * w), device=self.device, dtype=torch.float32)
        self._check(self.lib.e2v_vae_decode(self.ctx, z.contiguous().data_ptr(), b, f, h, w, int(postprocess),
                                            out.data_ptr(), _stream()))
        return out[:, :, 0] if squeeze else out

    def vae_encode(self, images: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        x = self._dev(images, "images")
        n, c, h, w = x.shape
        lat = self.vae_cfg.latent_channels
        mean = torch.empty((n, lat, h // 8, w // 8), device=self.device, dtype=torch.float32)
        logvar = torch.empty_like(mean)
        self._check(self.lib.e2v_vae_encode(self.ctx, x.data_ptr(), n, h, w, mean.data_ptr(), logvar.data_ptr(), _stream()))
        return mean, logvar

    def generate(self, latents: torch.Tensor, cond: torch.Tensor, uncond: Optional[torch.Tensor],
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, eta: float = 0.0, decode: bool = True,
                 return_latents: bool = False):
        x, cond = self._dev(latents, "latents"), self._dev(cond, "cond")
        un = self._dev(uncond, "uncond") if uncond is not None else None
        self._check_loop_inputs(x, cond, un)
        if guidance_scale > 1.0 and un is None:
            raise ValueError("classifier-free guidance (guidance_scale > 1) needs `uncond`")
        b, c, f, h, w = x.shape
        videos = torch.empty((b, self.vae_cfg.out_channels, f, 8 * h, 8 * w), device=self.device,
                             dtype=torch.float32) if decode else None
        lat_out = torch.empty_like(x) if return_latents else None
        self._check(self.lib.e2v_generate(
            self.ctx, x.data_ptr(), cond.data_ptr(), un.data_ptr() if un is not None else None,
            un.shape[0] if un is not None else 0, b, f, h, w, cond.shape[1], int(num_inference_steps),
            float(guidance_scale), float(eta), videos.data_ptr() if decode else None,
            lat_out.data_ptr() if return_latents else None, _stream()))
        return (videos, lat_out) if return_latents else videos

    # ------------------------------------------------------------------ the steps either side of the path
    def semantic_predict(self, eeg: torch.Tensor) -> torch.Tensor:
        x = self._dev(eeg, "eeg")
        if x.dim() != 2 or x.shape[1] != self.sem_cfg.in_features:
            raise ValueError(f"expected eeg [B,{self.sem_cfg.in_features}], got {tuple(x.shape)}")
        out = torch.empty((x.shape[0], self.sem_cfg.tokens * self.unet_cfg.cross_attention_dim), device=self.device,
                          dtype=torch.float32)
        self._check(self.lib.e2v_semantic_predict(self.ctx, x.data_ptr(), x.shape[0], out.data_ptr(), _stream()))
        return out

    def dana_noise(self, x0: torch.Tensor, eps_div: torch.Tensor, eps_same: torch.Tensor, t: Sequence[int],
                   dynamic_beta: float, time_steps: int = 500) -> torch.Tensor:
        """``[B,F,C,H,W]`` Seq2Seq latents + the two noise draws -> noised latents in the pipeline layout ``[B,C,F,H,W]``."""
        x0, ed, es = self._dev(x0, "x_0"), self._dev(eps_div, "eps_div"), self._dev(eps_same, "eps_same")
        b, f, c, h, w = x0.shape
        if tuple(ed.shape) != (b, f, c, h, w) or tuple(es.shape) != (b, 1, c, h, w):
            raise ValueError("noise shapes do not match x_0")
        ts = np.ascontiguousarray(np.asarray(t, dtype=np.int64).reshape(-1))
        if ts.size != b:
            raise ValueError("one timestep per clip is required")
        out = torch.empty((b, c, f, h, w), device=self.device, dtype=torch.float32)
        self._check(self.lib.e2v_dana_noise(self.ctx, x0.data_ptr(), ed.data_ptr(), es.data_ptr(),
                                            ts.ctypes.data_as(_lib.c_int64_p), int(time_steps), float(dynamic_beta), b, f, c,
                                            h, w, out.data_ptr(), _stream()))
        return out

    def frames_to_uint8(self, videos: torch.Tensor) -> torch.Tensor:
        v = self._dev(videos, "videos")
        out = torch.empty(v.shape, device=self.device, dtype=torch.uint8)
        self._check(self.lib.e2v_frames_to_uint8(self.ctx, v.data_ptr(), out.data_ptr(), v.numel(), _stream()))
        return out

    # ------------------------------------------------------------------ the one exchange of the path
    def comm_init(self, group=None) -> int:
        """Open the library's own RCCL communicator over the ranks of the (already initialised) ``torch.distributed`` group:
        rank 0 draws the id, ``torch.distributed`` only ships its 128 bytes.  Returns the world size."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        # one communicator per GROUP: a different group of the same size must not reuse it (its ranks are other processes)
        key = tuple(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
        if self.lib.e2v_comm_world(self.ctx) == world and getattr(self, "_comm_key", None) == key:
            return world
        if self.lib.e2v_comm_world(self.ctx) > 0:
            self.comm_destroy()
        buf = (C.c_ubyte * 128)()
        if rank == 0:
            if self.lib.e2v_comm_unique_id(buf) != _lib.E2V_OK:
                raise RuntimeError("e2v_comm_unique_id failed (is librccl.so loadable?)")
        ids = [bytes(buf)]
        dist.broadcast_object_list(ids, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = (C.c_ubyte * 128).from_buffer_copy(ids[0])
        self._check(self.lib.e2v_comm_init(self.ctx, raw, rank, world))
        self._comm_key = key
        return world

    def comm_destroy(self) -> None:
        self._check(self.lib.e2v_comm_destroy(self.ctx))
        self._comm_key = None

    def allgather_frames(self, frames: torch.Tensor, as_uint8: bool = False) -> torch.Tensor:
        """``[b,3,F,H,W]`` fp32 frames of this rank -> ``[world*b,3,F,H,W]`` of all ranks (fp32 or uint8), one ``ncclAllGather``
        issued by the library on the current stream; every rank must hold the same ``b``."""
        world = self.lib.e2v_comm_world(self.ctx)
        if world <= 0:
            raise RuntimeError("allgather_frames: no communicator (Engine.comm_init first)")
        v = self._dev(frames, "frames")
        out = torch.empty((world * v.shape[0],) + tuple(v.shape[1:]), device=self.device, dtype=torch.uint8 if as_uint8 else torch.float32)
        self._check(self.lib.e2v_allgather_frames(self.ctx, v.data_ptr(), v.numel(), int(as_uint8), out.data_ptr(), _stream()))
        return out

    # ------------------------------------------------------------------ kernel-level ops (channel-last tensors)
    def op_conv3x3(self, x0, w, bias=None, x1=None, *, n_img, Hs, Ws, Hi=None, Wi=None, stride=1, pad_lo=1, pad_hi=1,
                   rowbias=None, rows_per_sample=1, resid=None):
        Hi, Wi = Hi or Hs, Wi or Ws
        Ho = (Hi + pad_lo + pad_hi - 3) // stride + 1
        Wo = (Wi + pad_lo + pad_hi - 3) // stride + 1
        cout = w.shape[0]
        out = torch.empty((n_img * Ho * Wo, cout), device=self.device, dtype=torch.float32)
        p = lambda t: t.data_ptr() if t is not None else None
        self._check(self.lib.e2v_op_conv3x3(self.ctx, x0.data_ptr(), x0.shape[1], p(x1), x1.shape[1] if x1 is not None else 0,
                                            n_img, Hs, Ws, Hi, Wi, Ho, Wo, stride, pad_lo, w.data_ptr(), p(bias), cout,
                                            p(rowbias), rows_per_sample, p(resid), out.data_ptr(), _stream()))
        return out

    def op_linear(self, x, w, bias=None, resid=None, geglu=False):
        m, k = x.shape
        n = w.shape[0] // 2 if geglu else w.shape[0]
        out = torch.empty((m, n), device=self.device, dtype=torch.float32)
        p = lambda t: t.data_ptr() if t is not None else None
        self._check(self.lib.e2v_op_linear(self.ctx, x.data_ptr(), x.stride(0), m, k, w.data_ptr(), p(bias), n, p(resid),
                                           int(geglu), out.data_ptr(), _stream()))
        return out

    def op_rowblock_sums(self, x):
        """Canonical (sum, sum of squares) per 64-row block and column of ``x`` rounded to bf16: ``[rows / 64, C, 2]``."""
        rows, c = x.shape
        out = torch.empty((rows // 64, c, 2), device=self.device, dtype=torch.float32)
        self._check(self.lib.e2v_op_rowblock_sums(self.ctx, x.data_ptr(), rows, c, out.data_ptr(), _stream()))
        return out

    def op_groupnorm(self, x0, gamma, beta, *, samples, P, groups, eps, silu=False, x1=None):
        c = x0.shape[1] + (x1.shape[1] if x1 is not None else 0)
        out = torch.empty((samples * P, c), device=self.device, dtype=torch.float32)
        self._check(self.lib.e2v_op_groupnorm(self.ctx, x0.data_ptr(), x0.shape[1], x1.data_ptr() if x1 is not None else None,
                                              x1.shape[1] if x1 is not None else 0, samples, P, groups, float(eps),
                                              gamma.data_ptr(), beta.data_ptr(), int(silu), out.data_ptr(), _stream()))
        return out

    def op_layernorm(self, x, gamma, beta, eps=1e-5):
        out = torch.empty_like(x)
        self._check(self.lib.e2v_op_layernorm(self.ctx, x.data_ptr(), x.shape[0], x.shape[1], gamma.data_ptr(),
                                              beta.data_ptr(), float(eps), out.data_ptr(), _stream()))
        return out

    def op_attention(self, q, k, v, *, n, F, heads, D, Nq, Nk, mode, scale):
        """q/k/v: 2-D views (row stride = ``.stride(0)``) of channel-last buffers; returns [n*F*Nq, heads*D]."""
        out = torch.empty((n * F * Nq, heads * D), device=self.device, dtype=torch.float32)
        assert k.stride(0) == v.stride(0)
        self._check(self.lib.e2v_op_attention(self.ctx, q.data_ptr(), q.stride(0), k.data_ptr(), v.data_ptr(), k.stride(0),
                                              out.data_ptr(), out.stride(0), n, F, heads, D, Nq, Nk, mode, float(scale),
                                              _stream()))
        return out

    def op_temporal_attention(self, qkv, *, n, F, HW, heads, D, scale):
        out = torch.empty((n * F * HW, heads * D), device=self.device, dtype=torch.float32)
        self._check(self.lib.e2v_op_temporal_attention(self.ctx, qkv.data_ptr(), out.data_ptr(), n, F, HW, heads, D,
                                                       float(scale), _stream()))
        return out


def compute_dtype_code(dtype) -> int:
    """The ``e2v_dtype`` of an arithmetic mode named as a string or as the ``torch.dtype`` a caller of the reference passes."""
    names = {"fp32": _lib.E2V_F32, "f32": _lib.E2V_F32, "float32": _lib.E2V_F32, "bf16": _lib.E2V_BF16, "bfloat16": _lib.E2V_BF16,
             "fp16": _lib.E2V_F16, "f16": _lib.E2V_F16, "half": _lib.E2V_F16, "float16": _lib.E2V_F16, "f32x3": _lib.E2V_F32X3}
    if isinstance(dtype, torch.dtype):
        dtype = str(dtype).replace("torch.", "")
    if dtype not in names:
        raise ValueError(f"compute dtype {dtype!r}: one of {sorted(names)}")
    return names[dtype]


def describe_dispatch(dtype: str = "bf16", batch: int = 32, frames: int = 6, h: int = 36, w: int = 64, tokens: int = 77, config=None):
    """Which kernel and tile every launch of ``e2v_generate`` (one guided DDIM step + decode of ``batch`` clips) takes, as a list of
    ``"<launches>x <class> <shape> -> <kernel> <tile>"`` lines -- ``e2v_op_describe_dispatch`` on a host-only context: no GPU, no
    weights (``tests/test_dispatch_golden.py`` pins the SD-v1-4 table)."""
    lib = _lib.load()
    cfg = _lib.E2VConfig()
    lib.e2v_default_config(C.byref(cfg))
    if config is not None:
        for k, v in config.items():
            setattr(cfg, k, v)
    ctx = C.c_void_p()
    if lib.e2v_create(C.byref(cfg), -1, C.byref(ctx)) != _lib.E2V_OK:
        raise RuntimeError("e2v_create(host-only) failed")
    try:
        code = compute_dtype_code(dtype)
        need = C.c_int64(0)
        st = lib.e2v_op_describe_dispatch(ctx, code, batch, frames, h, w, tokens, None, 0, C.byref(need))
        if st != _lib.E2V_OK:
            raise RuntimeError(f"e2v_op_describe_dispatch: {(lib.e2v_last_error(ctx) or b'').decode()}")
        buf = C.create_string_buffer(need.value)
        st = lib.e2v_op_describe_dispatch(ctx, code, batch, frames, h, w, tokens, buf, need.value, C.byref(need))
        if st != _lib.E2V_OK:
            raise RuntimeError(f"e2v_op_describe_dispatch: {(lib.e2v_last_error(ctx) or b'').decode()}")
        return buf.value.decode().splitlines()
    finally:
        lib.e2v_destroy(ctx)
