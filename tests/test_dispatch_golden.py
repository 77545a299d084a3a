"""CPU: the dispatch of the SD-v1-4 UNet3D + VAE is pinned.  Which kernel a layer takes depends on M, N, K, taps, the tile count and the
residual (bgemm256.hip t256_tile_cols, bgemm.hip's launch rules, model.cpp Runner::winograd, attn_q64.hip's rule, norm.hip's chunk rows):
`e2v_op_describe_dispatch` walks e2v_generate as a dry run on a host-only context and records every launch's kernel and tile; the table
for B in {1, 8, 32} and both arithmetic modes is committed (tests/golden/dispatch_sd_v1_4.json, written by make_dispatch_golden.py), so a
rule change shows up HERE as a diff, not as a slower bench."""
import difflib
import json
import os

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dispatch_sd_v1_4.json")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("batch", [1, 8, 32])
def test_dispatch_table_of_sd_v1_4(dtype, batch):
    from eeg2video_amd.engine import describe_dispatch
    want = json.load(open(GOLDEN))[f"{dtype}_b{batch}"]
    got = describe_dispatch(dtype, batch)
    if got != want:
        diff = "\n".join(difflib.unified_diff(want, got, "golden", "this build", lineterm="", n=0))
        pytest.fail(f"dispatch of {dtype} B={batch} changed (deliberate? rerun tests/golden/make_dispatch_golden.py and commit the diff):\n{diff}")


def test_dispatch_table_covers_every_kernel_family():
    """Every class of the hot path appears with a kernel behind it, and the two modes really take different kernels."""
    table = json.load(open(GOLDEN))
    bf, fp = "\n".join(table["bf16_b32"]), "\n".join(table["fp32_b8"])
    for needle in ("bgemm_t256_kernel 256x320", "bgemm_t256p_kernel 256x256", "flash_attn_b16q64p_kernel w4", "cross_attn_resident_kernel",
                   "temporal_attn_wave_kernel", "gn_apply8_rows_kernel", "layernorm_bf16_rows_kernel"):
        assert needle in bf, needle
    for needle in ("igemm_k16_kernel 128x128x16", "igemm_kernel 128x128x32", "wino_in", "wino_out", "flash_attn_kernel", "layernorm_f32_rows_kernel"):
        assert needle in fp, needle
    assert "bgemm" not in fp and "wino" not in bf


def test_describe_needs_a_host_only_context_and_leaves_no_state():
    """Two descriptions of the same configuration are identical (the dry run keeps no state), and B changes the table (the tile-count rules)."""
    from eeg2video_amd.engine import describe_dispatch
    a, b = describe_dispatch("bf16", 1), describe_dispatch("bf16", 1)
    assert a == b and a != describe_dispatch("bf16", 32)


@pytest.mark.parametrize("batch", [1, 32])
def test_fp16_mode_takes_the_same_kernels_and_tiles_as_bf16(batch):
    """The fp16 mode (E2V_F16: the reference's own inference dtype) is the SAME launch rules over the other instance of every 16-bit
    kernel template (csrc/h16.h): its dispatch is the bf16 table with the two profile class names changed, launch for launch."""
    from eeg2video_amd.engine import describe_dispatch
    want = [l.replace("igemm_bf16", "igemm_fp16").replace("flash_attn_bf16", "flash_attn_fp16") for l in json.load(open(GOLDEN))[f"bf16_b{batch}"]]
    got = describe_dispatch("fp16", batch)
    if got != want:
        pytest.fail("fp16 dispatch differs from bf16's:\n" + "\n".join(difflib.unified_diff(want, got, "bf16 golden (renamed)", "fp16", lineterm="", n=0)))
