"""Sharding of clips over the GPUs of one node and the one collective of the path.

The reference generates clips in a serial ``for i in range(200)`` loop (``EEG2Video/inference_eeg2video.py:90``)
and no op of the UNet or VAE mixes samples, so clips partition over ranks with NO data-path exchange; the CFG
pair of a clip stays on one rank.  The only collective is the all-gather of the decoded frames at the end
(RCCL over xGMI on GPUs, ``backend="nccl"``; ``gloo`` in the CPU tests).  The node is fully connected, so the
library's all-gather runs as direct peer transfers; frames may be gathered as uint8 (4x fewer bytes) when the
caller only writes GIFs.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [start, stop) of ``total`` clips for ``rank`` (first ``total % world`` ranks get one more)."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def frames_to_uint8(frames: torch.Tensor) -> torch.Tensor:
    """[0,1] float frames -> uint8 as ``save_videos_grid`` does (``tuneavideo/util.py:29``: ``(x * 255).astype(uint8)``)."""
    return (frames * 255).to(torch.uint8)


def _shard_counts(n: int, device, group) -> list:
    """The clip count of every rank, exchanged ONCE per call: one stacked tensor, one host read."""
    world = dist.get_world_size(group)
    counts = torch.zeros(world, device=device, dtype=torch.int64)
    dist.all_gather_into_tensor(counts, torch.tensor([n], device=device, dtype=torch.int64), group=group)
    return counts.tolist()


def all_gather_frames(frames: torch.Tensor, group: Optional[dist.ProcessGroup] = None, as_uint8: bool = False,
                      force_collective: bool = False, engine=None, uniform: bool = False) -> torch.Tensor:
    """Every rank contributes ``[b_r, 3, F, H, W]`` (``b_r`` may differ by one between ranks) and receives the
    clips of all ranks in rank order, ``[sum b_r, 3, F, H, W]``.  Single-process: returns the input, unless
    ``force_collective`` asks for the collective to run even over a one-rank group (``bench.py --dist-single``: the RCCL
    code path on a one-GPU box).

    ``engine``: run the exchange BELOW the C ABI -- ``e2v_allgather_frames`` on the library's own RCCL communicator
    (``Engine.comm_init``; ``torch.distributed`` then only ships the 128-byte communicator id), with the uint8 conversion fused in
    front of it.  ``ncclAllGather`` needs the same ``b`` on every rank: the counts are exchanged first (ONE tiny all-gather and one
    host read per call, shared by both paths) and a ragged set of shards -- the last batch of a sweep -- goes through the padded
    ``torch.distributed`` path instead.

    ``uniform``: the caller KNOWS every rank holds the same ``b`` (the benchmark, the full batches of a sweep): no count exchange
    and no host synchronisation in front of the collective."""
    active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective)
    if not active:
        return frames_to_uint8(frames) if as_uint8 else frames
    world = dist.get_world_size(group)
    if uniform:
        sizes = [int(frames.shape[0])] * world
    else:
        dev = frames.device if frames.is_cuda or dist.get_backend(group) != "nccl" else torch.device("cuda")
        sizes = _shard_counts(int(frames.shape[0]), dev, group)
    same = len(set(sizes)) == 1
    if engine is not None and same:
        engine.comm_init(group)
        return engine.allgather_frames(frames, as_uint8=as_uint8)
    x = (frames_to_uint8(frames) if as_uint8 else frames).contiguous()
    bmax = max(sizes)
    if same:
        out = torch.empty((world * bmax,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
        dist.all_gather_into_tensor(out, x, group=group)
        return out
    pad = torch.zeros((bmax,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    pad[: x.shape[0]] = x
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)
