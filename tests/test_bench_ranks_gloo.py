"""CPU, world_size 2 over gloo: the RANK BODY of bench.py (`rank_body`) with a stub engine -- everything the N > 1 run does around
`e2v_generate` that a one-GPU box can never execute: per-rank seeds, the all-gather of the frames, rank 0's host buffer holding
world x B clips in rank order (fp32 and uint8 gathers), per-rank clocks, max-over-ranks timing and the JSON line's contract."""
import importlib.util
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FRAME = (3, 2, 4, 6)
LAT = (4, 2, 4, 6)
COND = (5, 8)
B = 3


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


class StubEngine:
    """Stands where the HIP engine stands in rank_body: clip k's "video" is a constant in [0, 1) derived from ITS latent and ITS
    conditioning, so that a buffer of gathered frames says which rank's which clip sits where."""
    device = torch.device("cpu")

    def __init__(self):
        self.calls, self.profiling = 0, False

    @staticmethod
    def clip_value(lat_k, cond_k):
        return float((lat_k.flatten()[0].abs() * 0.37 + cond_k.flatten()[0].abs() * 0.11) % 1.0)

    def generate(self, lat, cond, unc, steps, guidance, eta, decode=True, return_latents=False):
        self.calls += 1
        assert unc.shape == (1,) + COND and lat.shape[1:] == LAT and cond.shape[1:] == COND
        return torch.stack([torch.full(FRAME, self.clip_value(lat[k], cond[k])) for k in range(lat.shape[0])])

    def frames_to_uint8(self, f):
        return (f * 255).to(torch.uint8)

    def profile_begin(self):
        self.profiling = True

    def profile_end(self):
        self.profiling = False
        return {"igemm_f32": {"ms": 1.0, "launches": 10, "flops": 1e9, "bytes": 1e6}}

    def device_bytes(self):
        return 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, gather, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bench = _bench()
        args = bench.parse_args(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--backend", "gloo", "--gather", gather,
                                 "--ddim-steps", "3", "--no-cpu-baseline"])
        eng = StubEngine()
        r = bench.rank_body(args, rank, world, eng, use_dist=True, B=B, frame_shape=FRAME, latent_shape=LAT, cond_shape=COND,
                            sync=lambda: None, pin=False)
        lat, cond, _ = bench.synthetic_inputs(rank, B, "cpu", LAT, COND)
        q.put((rank, {"result": r["result"], "finite": r["finite"], "calls": eng.calls,
                      "host": None if r["host_frames"] is None else r["host_frames"].numpy(),
                      "lat0": float(lat[0].flatten()[0]), "values": [StubEngine.clip_value(lat[k], cond[k]) for k in range(B)]}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gather", ["fp32", "uint8"])
def test_rank_body_world_size_2(gather):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, gather, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = res[0], res[1]
    # the ranks generate DIFFERENT clips (seeds 1234 + rank * B + k)
    assert r0["lat0"] != r1["lat0"] and set(r0["values"]).isdisjoint(r1["values"])
    # only rank 0 prints a line and holds the host buffer; it holds world x B clips in rank order
    assert r1["result"] is None and r1["host"] is None and r0["finite"] and r1["finite"]
    host = r0["host"]
    assert host.shape == (2 * B,) + FRAME and host.dtype == (np.uint8 if gather == "uint8" else np.float32)
    want = r0["values"] + r1["values"]
    for k, v in enumerate(want):
        exp = np.uint8(np.float32(v) * np.float32(255)) if gather == "uint8" else np.float32(v)
        assert np.all(host[k] == exp), (k, host[k].flat[0], exp)
    # 1 warm-up (the instrumented one) + 2 timed + the one-step pass of the exchange timing, on both ranks
    assert r0["calls"] == 4 and r1["calls"] == 4
    d = r0["result"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "clips/s"
    assert d["config"]["clips_per_gpu"] == B and "model" not in d["config"] and "2xMI355X" in d["config"]["workload"]
    # whole-job value = all ranks' clips over the max-over-ranks time
    assert abs(d["value"] - 2 * B * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    pr = d["per_rank_clips_per_s"]
    assert len(pr["ms_per_step"]) == 2 and pr["min"] <= pr["max"] and all(m > 0 for m in pr["ms_per_step"])
    # a rank's own clock stops before the closing barrier: it cannot exceed the max-over-ranks figure by more than scheduling noise
    assert max(pr["ms_per_step"]) <= d["ms_per_step"] * 1.05 + 1.0
    assert d["gather"]["dtype"] == gather and d["gather_ms"] is not None and d["d2h_ms"] is not None
    assert d["rccl_ranks"] == 0            # gloo rehearsal: no RCCL rank is claimed
    assert d["roofline"]["kernel"] == "igemm_f32" and d["configs2"] is None and d["cpu_baseline"] is None
    assert "all-gather" in d["config"]["timed_region"] and "gloo" in d["config"]["collective"]
