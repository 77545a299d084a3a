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


def all_gather_frames(frames: torch.Tensor, group: Optional[dist.ProcessGroup] = None, as_uint8: bool = False,
                      force_collective: bool = False, engine=None) -> torch.Tensor:
    """Every rank contributes ``[b_r, 3, F, H, W]`` (``b_r`` may differ by one between ranks) and receives the
    clips of all ranks in rank order, ``[sum b_r, 3, F, H, W]``.  Single-process: returns the input, unless
    ``force_collective`` asks for the collective to run even over a one-rank group (``bench.py --dist-single``: the RCCL
    code path on a one-GPU box).

    ``engine``: run the exchange BELOW the C ABI -- ``e2v_allgather_frames`` on the library's own RCCL communicator
    (``Engine.comm_init``; ``torch.distributed`` then only ships the 128-byte communicator id), with the uint8 conversion fused in
    front of it.  ``ncclAllGather`` needs the same ``b`` on every rank: the counts are exchanged first (one tiny all-gather) and a
    ragged set of shards -- the last batch of a sweep -- goes through the padded ``torch.distributed`` path below instead."""
    if engine is not None and dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective):
        world = dist.get_world_size(group)
        dev = frames.device if frames.is_cuda or dist.get_backend(group) != "nccl" else torch.device("cuda")
        counts = torch.tensor([frames.shape[0]], device=dev, dtype=torch.int64)
        all_counts = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(all_counts, counts, group=group)
        if len({int(c.item()) for c in all_counts}) == 1:
            engine.comm_init(group)
            return engine.allgather_frames(frames, as_uint8=as_uint8)
    x = frames_to_uint8(frames) if as_uint8 else frames
    if not (dist.is_available() and dist.is_initialized()):
        return x
    if dist.get_world_size(group) == 1 and not force_collective:
        return x
    world = dist.get_world_size(group)
    counts = torch.tensor([x.shape[0]], device=x.device, dtype=torch.int64)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    sizes = [int(c.item()) for c in all_counts]
    bmax = max(sizes)
    x = x.contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((world * bmax,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
        dist.all_gather_into_tensor(out, x, group=group)
        return out
    pad = torch.zeros((bmax,) + tuple(x.shape[1:]), device=x.device, dtype=x.dtype)
    pad[: x.shape[0]] = x
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)
