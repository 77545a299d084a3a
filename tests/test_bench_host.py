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
