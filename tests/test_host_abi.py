"""CPU: the C-ABI library loads, exports every symbol the headers declare, and its host-side parts
(state-dict key scheme, DDIM schedule) agree with the oracle and the reference-derived spec.  No GPU work."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from eeg2video_amd import _lib
from eeg2video_amd.weights import (SemanticConfig, UNetConfig, VAEConfig, semantic_param_spec, unet_param_spec,
                                   vae_param_spec)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


@pytest.fixture(scope="module")
def host_ctx(lib):
    cfg = _lib.E2VConfig()
    lib.e2v_default_config(C.byref(cfg))
    ctx = C.c_void_p()
    assert lib.e2v_create(C.byref(cfg), -1, C.byref(ctx)) == 0
    yield ctx
    lib.e2v_destroy(ctx)


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(e2v_[a-z0-9_]+)\s*\(", text))


def test_every_declared_symbol_is_exported_and_bound(lib):
    declared = _declared("eeg2video_hip.h") | _declared("eeg2video_hip_ops.h")
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None


def test_default_config_is_sd_v1_4(lib):
    cfg = _lib.E2VConfig()
    lib.e2v_default_config(C.byref(cfg))
    assert list(cfg.block_out_channels) == [320, 640, 1280, 1280] and cfg.cross_attention_dim == 768
    assert cfg.attention_heads == 8 and cfg.norm_num_groups == 32 and abs(cfg.norm_eps - 1e-5) < 1e-12
    assert list(cfg.vae_block_out_channels) == [128, 256, 512, 512] and cfg.vae_scaling_factor == 0.18215
    assert cfg.num_train_timesteps == 1000 and cfg.steps_offset == 1 and cfg.beta_start == 0.00085


def test_key_scheme_matches_python_spec(lib, host_ctx):
    n = lib.e2v_num_expected_keys(host_ctx)
    shape, nd = (C.c_int64 * 4)(), C.c_int()
    got = {}
    for i in range(n):
        k = lib.e2v_expected_key(host_ctx, i, shape, C.byref(nd)).decode()
        got[k] = tuple(shape[d] for d in range(nd.value))
    want = dict(unet_param_spec(UNetConfig()))
    want.update({"vae." + k: v for k, v in vae_param_spec(VAEConfig()).items()})
    want.update({"semantic." + k: v for k, v in semantic_param_spec(SemanticConfig(), 768).items()})
    assert got == want and n == 798 + 248 + 10


def test_ddim_timesteps_bit_exact(lib, host_ctx):
    from oracle import DDIMOracle
    for n in (1, 2, 3, 4, 7, 20, 50, 100, 250, 333, 1000):
        out = np.empty(n, dtype=np.int64)
        assert lib.e2v_ddim_timesteps(host_ctx, n, out.ctypes.data_as(_lib.c_int64_p)) == 0
        assert np.array_equal(out, DDIMOracle().set_timesteps(n)), n
    out = np.empty(4, dtype=np.int64)
    lib.e2v_ddim_timesteps(host_ctx, 4, out.ctypes.data_as(_lib.c_int64_p))
    assert out.tolist() == [751, 501, 251, 1]
    assert lib.e2v_ddim_timesteps(host_ctx, 0, out.ctypes.data_as(_lib.c_int64_p)) == _lib.E2V_EINVAL


def test_alpha_table_matches_torch_to_the_last_bits(lib, host_ctx):
    from oracle import DDIMOracle
    a = np.empty(1000, dtype=np.float32)
    assert lib.e2v_ddim_alphas_cumprod(host_ctx, a.ctypes.data_as(C.POINTER(C.c_float))) == 0
    ref = DDIMOracle().alphas_cumprod.numpy()
    assert np.max(np.abs(a - ref) / ref) < 1e-5
    ref2 = ref.copy()
    assert lib.e2v_set_alphas_cumprod(host_ctx, ref2.ctypes.data_as(C.POINTER(C.c_float)), 1000) == 0
    lib.e2v_ddim_alphas_cumprod(host_ctx, a.ctypes.data_as(C.POINTER(C.c_float)))
    assert np.array_equal(a, ref)


def test_set_ddim_schedule_moves_offset_and_table(lib):
    """DDIMScheduler.bind pushes (table, steps_offset) so that the fused loop and a stepped loop agree."""
    cfg = _lib.E2VConfig()
    lib.e2v_default_config(C.byref(cfg))
    ctx = C.c_void_p()
    assert lib.e2v_create(C.byref(cfg), -1, C.byref(ctx)) == 0
    tab = np.linspace(0.999, 0.01, 500).astype(np.float32)
    assert lib.e2v_set_ddim_schedule(ctx, tab.ctypes.data_as(C.POINTER(C.c_float)), 500, 0) == 0
    out = np.empty(5, dtype=np.int64)
    assert lib.e2v_ddim_timesteps(ctx, 5, out.ctypes.data_as(_lib.c_int64_p)) == 0
    assert out.tolist() == [400, 300, 200, 100, 0]
    got = np.empty(500, dtype=np.float32)
    assert lib.e2v_ddim_alphas_cumprod(ctx, got.ctypes.data_as(C.POINTER(C.c_float))) == 0 and np.array_equal(got, tab)
    assert lib.e2v_set_ddim_schedule(ctx, tab.ctypes.data_as(C.POINTER(C.c_float)), 500, 500) == _lib.E2V_EINVAL
    lib.e2v_destroy(ctx)


def test_create_rejects_head_dims_without_a_kernel(lib):
    """block_out_channels / heads must be a head dim flash_attention has an instance for: D = 48 used to be accepted
    and ran the UNet with an uninitialised attention output."""
    cfg = _lib.E2VConfig()
    lib.e2v_default_config(C.byref(cfg))
    cfg.block_out_channels = (C.c_int * 4)(384, 640, 1280, 1280)       # 384 / 8 = 48
    ctx = C.c_void_p()
    assert lib.e2v_create(C.byref(cfg), -1, C.byref(ctx)) == _lib.E2V_EINVAL
    assert b"head dim" in lib.e2v_last_error(None)
    cfg.block_out_channels = (C.c_int * 4)(320, 640, 1280, 1280)
    assert lib.e2v_create(C.byref(cfg), -1, C.byref(ctx)) == 0
    lib.e2v_destroy(ctx)


def test_run_time_switches_by_name(lib):
    """e2v_op_set_knob: the switches of DESIGN section 10 by the name of their environment variable; unknown names are refused."""
    assert lib.e2v_op_set_knob(b"E2V_BGEMM_PERS", 0) == _lib.E2V_OK
    assert lib.e2v_op_set_knob(b"E2V_BGEMM_PERS", 1) == _lib.E2V_OK
    assert lib.e2v_op_set_knob(b"E2V_BGEMM_256", 1) == _lib.E2V_OK
    assert lib.e2v_op_set_knob(b"E2V_NO_SUCH_SWITCH", 1) == _lib.E2V_EINVAL
    assert lib.e2v_op_set_knob(None, 1) == _lib.E2V_EINVAL


def test_device_entry_points_refuse_host_only_context(lib, host_ctx):
    assert lib.e2v_finalize_weights(host_ctx, 1) == _lib.E2V_ESTATE
    assert b"host-only" in lib.e2v_last_error(host_ctx)
    x = np.zeros(4, np.float32)
    shape = (C.c_int64 * 1)(4)
    assert lib.e2v_load_tensor(host_ctx, b"conv_in.bias", x.ctypes.data_as(C.c_void_p), 0, shape, 1) == _lib.E2V_ESTATE


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from eeg2video_amd.engine import Engine
    with pytest.raises(RuntimeError, match="no CPU path"):
        Engine()


def test_mirror_input_checks_need_no_gpu():
    """check_inputs / prepare_latents raise the reference's ValueErrors before any device work."""
    import torch
    from eeg2video_amd.pipeline import TuneAVideoPipeline
    p = object.__new__(TuneAVideoPipeline)
    p.vae_scale_factor = 8
    with pytest.raises(ValueError, match="has to be of type"):
        p.check_inputs("eeg", 288, 512, 1)
    with pytest.raises(ValueError, match="divisible by 8"):
        p.check_inputs(torch.zeros(1), 290, 512, 1)
    with pytest.raises(ValueError, match="positive integer"):
        p.check_inputs(torch.zeros(1), 288, 512, None)
    with pytest.raises(ValueError, match="Unexpected latents shape"):
        p.prepare_latents(1, 4, 6, 288, 512, torch.float32, torch.device("cpu"), None, torch.zeros(1, 4, 6, 36, 63))


def test_config_struct_size_is_guarded_across_the_abi(lib):
    """e2v_config_size(): the library's sizeof(e2v_config) equals the binding's, and the struct a maintainer would paste from
    INTEGRATION.md section 2 (extracted from the document and executed) has that size too -- a stale field list fails here instead
    of e2v_default_config writing past the caller's object."""
    assert lib.e2v_config_size() == C.sizeof(_lib.E2VConfig)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"^class E2VConfig\(C\.Structure\):.*?\n(?=\n)", text, flags=re.S | re.M)
    assert m, "INTEGRATION.md no longer shows the E2VConfig binding"
    ns = {"C": C}
    exec(m.group(0), ns)
    doc_cfg = ns["E2VConfig"]
    assert C.sizeof(doc_cfg) == lib.e2v_config_size()
    assert [f[0] for f in doc_cfg._fields_] == [f[0] for f in _lib.E2VConfig._fields_]
    cfg = doc_cfg()
    lib.e2v_default_config(C.cast(C.byref(cfg), C.POINTER(_lib.E2VConfig)))
    assert cfg.sem_in_features == 310 and cfg.sem_hidden == 10000 and cfg.sem_tokens == 77 and cfg.steps_offset == 1


def test_collective_entry_points_report_call_order_errors(lib, host_ctx):
    """The C-ABI collective without a GPU: no communicator -> world 0, E2V_ESTATE from the gather (host-only context), destroy of
    nothing is fine, null arguments are E2V_EINVAL.  (The RCCL calls themselves run in the -m gpu test.)"""
    assert lib.e2v_comm_world(host_ctx) == 0
    assert lib.e2v_comm_destroy(host_ctx) == 0
    buf = (C.c_float * 4)()
    assert lib.e2v_allgather_frames(host_ctx, buf, 4, 0, buf, None) == _lib.E2V_ESTATE
    assert lib.e2v_allgather_frames(None, buf, 4, 0, buf, None) == _lib.E2V_EINVAL
    assert lib.e2v_comm_init(None, buf, 0, 1) == _lib.E2V_EINVAL
    assert lib.e2v_comm_unique_id(None) == _lib.E2V_EINVAL


def test_scheduler_config_with_unsupported_arithmetic_is_refused():
    """scheduler_from_config: a key that changes the arithmetic and is set to something this build does not implement must raise,
    whether or not the mirror's constructor knows the key -- a v_prediction DDIM config (diffusers 0.11.1 supports it) must not
    load silently as epsilon prediction.  Descriptive keys are dropped; clip_sample = True is overridden as the pipeline's
    constructor does (pipeline_tuneeeg2video.py:73-84)."""
    from eeg2video_amd.scheduler import scheduler_from_config
    base = {"_class_name": "DDIMScheduler", "_diffusers_version": "0.11.1", "beta_start": 0.00085, "beta_end": 0.012,
            "beta_schedule": "scaled_linear", "num_train_timesteps": 1000, "steps_offset": 1, "trained_betas": None}
    s = scheduler_from_config(dict(base, clip_sample=True, prediction_type="epsilon"))
    assert type(s).__name__ == "DDIMScheduler" and s.config.clip_sample is False
    for cls in ("DDIMScheduler", "PNDMScheduler", "EulerDiscreteScheduler", "DPMSolverMultistepScheduler"):
        with pytest.raises(NotImplementedError, match="prediction_type"):
            scheduler_from_config(dict(base, _class_name=cls, prediction_type="v_prediction"))
    with pytest.raises(NotImplementedError, match="trained_betas"):
        scheduler_from_config(dict(base, trained_betas=[0.1, 0.2]))
    with pytest.raises(NotImplementedError, match="set_alpha_to_one"):
        scheduler_from_config(dict(base, set_alpha_to_one=True))
    # timestep_spacing: per class, the spacing that class implements (newer diffusers write the default into the config)
    for cls, ok, bad in (("DDIMScheduler", "leading", "linspace"), ("PNDMScheduler", "leading", "trailing"),
                         ("EulerDiscreteScheduler", "linspace", "leading"), ("EulerAncestralDiscreteScheduler", "linspace", "trailing"),
                         ("LMSDiscreteScheduler", "linspace", "leading"), ("DPMSolverMultistepScheduler", "linspace", "leading")):
        assert type(scheduler_from_config(dict(base, _class_name=cls, timestep_spacing=ok))).__name__ == cls
        with pytest.raises(NotImplementedError, match="timestep_spacing"):
            scheduler_from_config(dict(base, _class_name=cls, timestep_spacing=bad))
    with pytest.raises(NotImplementedError, match="thresholding"):
        scheduler_from_config(dict(base, _class_name="DPMSolverMultistepScheduler", thresholding=True))
    with pytest.raises(ValueError):
        scheduler_from_config(dict(base, _class_name="KarrasVeScheduler"))
