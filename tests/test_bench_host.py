"""CPU: the host-side logic of bench.py -- the self-launch of an N-rank job (the parent never touches the GPU and never
re-execs: it starts children), the CPU-share detection of the CPU baseline."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_importing_bench_does_not_import_torch_cuda_state(bench):
    """The parent of a self-launch must be able to decide and spawn before anything initialises the GPU: the module
    imports no torch at top level."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src.split("def main()")[0]
    assert "\nimport torch" not in head and "\nfrom torch" not in head


def test_self_launch_command(bench, monkeypatch):
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    assert bench.main() == 7                                   # the children's exit code is passed on
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_rank_count_mismatch_is_refused(bench, monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    assert bench.main() == 2


def test_host_cores_is_capped_by_affinity(bench):
    hc = bench.host_cores()
    assert 1 <= hc["cores"] <= hc["affinity"] <= (hc["os_cpu_count"] or 10**6)
    assert isinstance(hc["cpu_model"], str)


def test_gather_options_and_budget_guard_parse(bench):
    a = bench.parse_args(["--gpus", "8", "--dtype", "bf16", "--batch", "32", "--gather", "uint8", "--gather-impl", "cabi"])
    assert (a.gpus, a.dtype, a.batch, a.gather, a.gather_impl) == (8, "bf16", 32, "uint8", "cabi")
    d = bench.parse_args([])
    assert d.gather == "fp32" and d.gather_impl == "torch" and not d.no_configs2 and 300 < d.configs2_budget_s < 600
    assert 0 <= bench.process_age_s() < 3600


def test_committed_bench_lines_keep_the_contract_and_quote_the_committed_pmc_traffic():
    """profiles/r05_bench_*.json are the lines `bench.py` printed on the GPU box: the driver's contract keys, the `roofline` and
    `cpu_baseline` objects, the `configs2` leg in the default line (4 timed passes) -- and `roofline.traffic` must be the figure of
    profiles/pmc_dominant_kernel.json for that batch / dtype (the PMC passes and the bench lines are regenerated together)."""
    import json
    prof = os.path.join(ROOT, "profiles")
    pmc = json.load(open(os.path.join(prof, "pmc_dominant_kernel.json")))
    d = json.load(open(os.path.join(prof, "r05_bench_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] and d["unit"] == "clips/s" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["dtype"] == "f32"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["kernel"] == "igemm_f32" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] == pmc["igemm_f32"]["hbm_bytes_per_launch"] and pmc["igemm_f32"]["batch"] == 8
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    assert d["parity"]["frames_max_abs"] < d["parity"]["tolerance_frames_max_abs"]
    c2 = d["configs2"]
    assert c2["dtype"] == "bf16" and c2["value"] > 0 and c2["steps"] == 4 and c2["roofline"]["kernel"] == "igemm_bf16"
    assert c2["roofline"]["traffic"] == pmc["igemm_bf16"]["hbm_bytes_per_launch"] and pmc["igemm_bf16"]["batch"] == 32
    b = json.load(open(os.path.join(prof, "r05_bench_bf16_b32.json")))
    assert b["dtype"] == "bf16" and b["roofline"]["traffic"] == pmc["igemm_bf16"]["hbm_bytes_per_launch"]
    h = json.load(open(os.path.join(prof, "r05_bench_fp16_b32.json")))     # the fp16 mode's line: the same contract, its own class and tolerance
    assert h["dtype"] == "f16" and h["roofline"]["kernel"] == "igemm_fp16" and h["roofline"]["peak"] == b["roofline"]["peak"]
    assert h["roofline"]["traffic"] == pmc["igemm_fp16"]["hbm_bytes_per_launch"] and pmc["igemm_fp16"]["dtype"] == "fp16"
    assert h["parity"]["frames_max_abs"] < h["parity"]["tolerance_frames_max_abs"] <= 0.25 * b["parity"]["tolerance_frames_max_abs"]
    assert 8.0 * h["parity"]["frames_max_abs"] < 1.2 * b["parity"]["frames_max_abs"]      # an eighth of the bf16 distance (three more mantissa bits)
    # README / DESIGN quote the 16-bit parity figures of THESE committed lines (they move with every change of a rounding)
    for line in (b, h):
        quoted = f"{line['parity']['frames_max_abs']:.1e}".replace("e-0", "e-")
        for doc in ("README.md", "DESIGN.md"):
            assert quoted in open(os.path.join(ROOT, doc)).read(), (doc, quoted)
    wp = b["roofline"]["whole_path"]
    assert abs(wp["achieved_tflops"] - b["value"] * wp["algorithmic_tflop_per_clip"]) < 1e-6 and abs(wp["frac"] - wp["achieved_tflops"] / wp["peak"]) < 1e-9
