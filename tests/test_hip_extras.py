"""GPU: the steps either side of the hot path through the C ABI, against the oracle."""
import os

import numpy as np
import pytest
import torch

from eeg2video_amd.weights import (TINY_SEMANTIC, TINY_UNET, TINY_VAE, counter_normal, counter_uniform, semantic_param_spec,
                                   synth_state_dict)

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


@pytest.fixture(scope="module")
def eng():
    from eeg2video_amd.engine import Engine
    return Engine(TINY_UNET, TINY_VAE, 0, sem_cfg=TINY_SEMANTIC)


@pytest.mark.parametrize("batch", [1, 5, 130])
def test_semantic_predictor_vs_oracle(eng, batch):
    """in_features = 22 is not a multiple of 4 (like the reference's 310): exercises the K padding."""
    from eeg2video_amd.semantic import CLIP
    from oracle import semantic_predictor
    spec = semantic_param_spec(TINY_SEMANTIC, TINY_UNET.cross_attention_dim)
    sd = synth_state_dict(spec, seed=44, mode="perturbed")
    model = CLIP(TINY_SEMANTIC, engine=eng).load_state_dict({"state_dict": sd})
    eeg = _t(counter_normal(3, "eeg", (batch, TINY_SEMANTIC.in_features)))
    ref = semantic_predictor({k: _t(v) for k, v in sd.items()}, eeg)
    out = model(eeg.cuda())
    assert out.shape == (batch, TINY_SEMANTIC.tokens * TINY_UNET.cross_attention_dim)
    err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-5, err


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


@pytest.mark.parametrize("batch", [1, 130])
def test_semantic_predictor_bf16_mode(eng, batch):
    """bf16 arithmetic (e2v_set_compute_dtype): the first layer's K is padded (22 -> 24) and every weight form must share
    that row length (the bf16 copy used to keep the unpadded stride).  Checked against the oracle with
    operands rounded to bf16 per layer (fp32 accumulate), and loosely against the fp32 oracle."""
    from eeg2video_amd.semantic import CLIP
    from oracle import semantic_predictor
    spec = semantic_param_spec(TINY_SEMANTIC, TINY_UNET.cross_attention_dim)
    sd = synth_state_dict(spec, seed=44, mode="perturbed")
    model = CLIP(TINY_SEMANTIC, engine=eng).load_state_dict({"state_dict": sd})
    eeg = _t(counter_normal(3, "eeg", (batch, TINY_SEMANTIC.in_features)))
    x = eeg
    for i in range(5):
        w, b = _t(sd[f"mlp.{2 * i}.weight"]), _t(sd[f"mlp.{2 * i}.bias"])
        x = _bf16(x) @ _bf16(w).T + b
        if i < 4:
            x = x.relu()
    ref32 = semantic_predictor({k: _t(v) for k, v in sd.items()}, eeg)
    eng.set_compute_dtype("bf16")
    try:
        out = model(eeg.cuda()).cpu()
    finally:
        eng.set_compute_dtype("fp32")
    scale = ref32.abs().max().item()
    assert (out - x).abs().max().item() / scale < 2e-3
    assert (out - ref32).abs().max().item() / scale < 5e-2


def test_dana_noise_vs_oracle(eng):
    from oracle import dana_noise
    b, f, c, h, w = 3, 6, 4, 9, 8
    x0 = _t(counter_normal(1, "x0", (b, f, c, h, w)))
    ed = _t(counter_normal(2, "ed", (b, f, c, h, w)))
    es = _t(counter_normal(3, "es", (b, 1, c, h, w)))
    t = torch.tensor([0, 137, 499])
    for beta in (0.3, 0.2):                                 # DANA's two dynamic_beta values (by optical-flow label)
        ref = dana_noise(x0, ed, es, t, beta)
        out = eng.dana_noise(x0.cuda(), ed.cuda(), es.cuda(), t.tolist(), beta)
        assert out.shape == (b, c, f, h, w)
        assert (out.cpu() - ref).abs().max().item() < 2e-6
    with pytest.raises(ValueError):
        eng.dana_noise(x0.cuda(), ed.cuda(), es.cuda(), [0, 1, 500], 0.3)


def test_frames_to_uint8_bit_exact(eng):
    from oracle import frames_to_uint8
    v = _t(counter_uniform(5, "v", 2 * 3 * 2 * 17 * 9)).reshape(2, 3, 2, 17, 9)
    v[0, 0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.999999, 1.0 / 255.0])
    out = eng.frames_to_uint8(v.cuda())
    assert out.dtype == torch.uint8 and np.array_equal(out.cpu().numpy(), frames_to_uint8(v))


def test_end_to_end_example_runs_on_the_tiny_configuration(monkeypatch):
    """examples/inference_eeg2video.py: semantic predictor -> DANA noise -> TuneAVideoPipeline.__call__ -> uint8 frames, the flow
    of the reference's inference script, on the tiny configuration."""
    import importlib.util, os, sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "inference_eeg2video.py")
    spec = importlib.util.spec_from_file_location("e2v_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["inference_eeg2video.py", "--tiny", "--steps", "3", "--clips", "2"])
    video = mod.main()
    assert tuple(video.shape) == (2, 3, 3, 32, 48)


def test_configs4_sweep_driver_on_the_tiny_configuration(tmp_path, capsys):
    """examples/run_sweep.py (BASELINE configs[4]): GLMNet + Seq2Seq host models -> semantic predictor -> DANA -> e2v_generate ->
    uint8 -> GIF writer over concepts x clips, batched with a ragged last batch; the same clips in one batch and in batches of 3
    agree (the HIP path is bit-identical per clip; the host torch transformer's GEMMs may round differently with the batch
    size, so the frames are compared to two uint8 levels); in the bf16-activation mode it merely runs."""
    import importlib.util, json, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "run_sweep.py")
    spec = importlib.util.spec_from_file_location("e2v_sweep", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    a = mod.main(["--tiny", "--concepts", "2", "--per-concept", "2", "--batch", "3", "--steps", "2", "--out", str(tmp_path), "--npy"])
    rec = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rec["clips"] == 4 and rec["clips_per_s"] > 0 and 0 <= rec["host_model_share"] < 1
    assert tuple(a.shape) == (4, 3, 3, 32, 48) and a.dtype == torch.uint8
    assert sorted(os.listdir(tmp_path)) == ["00_0.npy", "00_1.npy", "01_0.npy", "01_1.npy"]
    assert np.array_equal(np.load(tmp_path / "01_0.npy")[:, :, :, :], a[2].permute(1, 2, 3, 0).numpy())
    b = mod.main(["--tiny", "--concepts", "2", "--per-concept", "2", "--batch", "4", "--steps", "2"])
    assert (a.int() - b.int()).abs().max().item() <= 2
    c = mod.main(["--tiny", "--concepts", "1", "--per-concept", "2", "--batch", "2", "--steps", "2", "--dtype", "bf16"])
    assert tuple(c.shape) == (2, 3, 3, 32, 48)


def test_cabi_allgather_frames_world_size_1(eng):
    """e2v_allgather_frames: the exchange of the decoded frames below the C ABI, on the library's own RCCL
    communicator.  One GPU per box here, so the world is one rank -- that still runs ncclGetUniqueId / ncclCommInitRank /
    ncclAllGather on the device: fp32 gather = the input, uint8 gather = bit-exact (x * 255) truncation; call-order errors are
    reported, not crashed on.  torch.distributed (gloo) only ships the 128-byte id."""
    import ctypes as C
    import torch.distributed as dist
    from eeg2video_amd import _lib
    from eeg2video_amd.dist import all_gather_frames, frames_to_uint8
    v = torch.rand(2, 3, 3, 16, 24, generator=torch.Generator().manual_seed(3)).cuda()
    with pytest.raises(RuntimeError, match="no communicator"):
        eng.allgather_frames(v)
    assert eng.lib.e2v_allgather_frames(eng.ctx, v.data_ptr(), v.numel(), 0, v.data_ptr(), None) == _lib.E2V_ESTATE
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        assert eng.comm_init() == 1 and eng.lib.e2v_comm_world(eng.ctx) == 1
        buf = (C.c_ubyte * 128)()
        assert eng.lib.e2v_comm_init(eng.ctx, buf, 0, 1) == _lib.E2V_ESTATE        # already holds one
        out = eng.allgather_frames(v)
        u8 = eng.allgather_frames(v, as_uint8=True)
        torch.cuda.synchronize()
        assert out.dtype == torch.float32 and torch.equal(out, v)
        assert u8.dtype == torch.uint8 and torch.equal(u8.cpu(), frames_to_uint8(v.cpu()))
        # the same through the dist helper bench.py / run_sweep.py call
        assert torch.equal(all_gather_frames(v, force_collective=True, engine=eng), v)
        assert torch.equal(all_gather_frames(v, as_uint8=True, force_collective=True, engine=eng), u8)
        eng.comm_destroy()
        assert eng.lib.e2v_comm_world(eng.ctx) == 0
    finally:
        dist.destroy_process_group()
