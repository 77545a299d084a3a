"""CPU, world_size 2 over gloo: the N > 1 path of the benchmark / inference driver -- shard the clips, no
data-path collective, one all-gather of the frames at the end, max-over-ranks timing."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from eeg2video_amd.dist import all_gather_frames, frames_to_uint8, shard_range


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 8, 200, 256):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_single_process_gather_is_identity():
    x = torch.rand(2, 3, 2, 4, 4)
    assert all_gather_frames(x) is x
    assert all_gather_frames(x, as_uint8=True).dtype == torch.uint8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(total, rank, world)
        # "generate" the rank's clips: clip k is a constant k/total video
        mine = torch.stack([torch.full((3, 2, 4, 4), k / total) for k in range(lo, hi)]) if hi > lo else torch.zeros(0, 3, 2, 4, 4)
        allf = all_gather_frames(mine)
        ok = allf.shape[0] == total and all(torch.all(allf[k] == k / total) for k in range(total))
        u8 = all_gather_frames(mine, as_uint8=True)
        ok = ok and torch.equal(u8, frames_to_uint8(allf))
        # max-over-ranks timing as bench.py does it
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = ok and t.item() == float(world)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 5])
def test_world_size_2_shard_and_gather(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}
