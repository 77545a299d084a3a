"""Regenerate tests/golden/dispatch_sd_v1_4.json: for the SD-v1-4 UNet3D + VAE at B in {1, 8, 32} clips and both arithmetic modes, the
kernel and tile every launch of one guided DDIM step + decode takes (e2v_op_describe_dispatch: a dry run of e2v_generate on a host-only
context -- no GPU needed).  Run it after a DELIBERATE change of a launch rule and commit the diff; tests/test_dispatch_golden.py
asserts the table, so an accidental change shows up as a failing CPU test instead of a slower bench.

    python tests/golden/make_dispatch_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def describe_all():
    from eeg2video_amd.engine import describe_dispatch
    out = {}
    for dtype in ("fp32", "bf16"):
        for b in (1, 8, 32):
            out[f"{dtype}_b{b}"] = describe_dispatch(dtype, b)
    return out


if __name__ == "__main__":
    table = describe_all()
    path = os.path.join(ROOT, "tests", "golden", "dispatch_sd_v1_4.json")
    json.dump({"_source": "tests/golden/make_dispatch_golden.py (e2v_op_describe_dispatch, SD-v1-4 config, F=6, 36x64 latents, 77 tokens, "
                          "one guided DDIM step + VAE decode); line = '<launches>x <class> <shape> -> <kernel> <tile>'", **table},
              open(path, "w"), indent=0)
    print(path, {k: len(v) for k, v in table.items()})
