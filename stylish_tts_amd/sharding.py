"""Utterance sharding across the GPUs of one node.

Utterances are independent units (no cross-utterance state in the reference; the shared ``rand(1,1)`` initial phase
of ``models/generator.py:306`` is passed explicitly per shard), so the path shards with NO data-path collective:
  * once: broadcast of the weights from rank 0 (RCCL over xGMI; ``gloo`` in CPU tests),
  * per batch: gather of the variable-length waveforms to rank 0.
Partitioning is longest-first greedy bin packing of frame counts (cost is linear in frames), ties by index.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def partition_utterances(frame_counts: Sequence[int], world: int) -> List[List[int]]:
    """Greedy longest-processing-time assignment; returns, per rank, the utterance indices (ascending)."""
    order = sorted(range(len(frame_counts)), key=lambda i: (-int(frame_counts[i]), i))
    load = [0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += int(frame_counts[i])
    return [sorted(p) for p in parts]


def broadcast_state_dict(sd: Optional[Dict[str, np.ndarray]], spec, device, src: int = 0) -> "OrderedDict[str, np.ndarray]":
    """One flat fp32 broadcast of every tensor named in `spec` (list of (name, shape, kind)); every rank returns
    the same state dict.  `sd` is only read on `src`."""
    total = int(sum(int(np.prod(s)) for _, s, _ in spec))
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        host = np.concatenate([np.asarray(sd[n], np.float32).reshape(-1) for n, _, _ in spec])
        flat.copy_(torch.from_numpy(host))
    dist.broadcast(flat, src=src)
    host = flat.cpu().numpy()
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    off = 0
    for n, s, _ in spec:
        k = int(np.prod(s))
        out[n] = host[off : off + k].reshape(s).copy()
        off += k
    return out


def gather_waveforms(local_audio: torch.Tensor, local_ids: Sequence[int], sample_counts: Sequence[int], dst: int = 0):
    """Gather each rank's concatenated waveforms to `dst` and return them in global utterance order
    (list of 1-D tensors on `dst`, None elsewhere).  sample_counts: samples of EVERY utterance (global order);
    the partition is recomputed identically on every rank, so only payload moves."""
    world, rank = dist.get_world_size(), dist.get_rank()
    parts = partition_utterances([int(c) for c in sample_counts], world)
    assert list(local_ids) == parts[rank], "local utterance ids do not match the deterministic partition"
    sizes = [int(sum(sample_counts[i] for i in p)) for p in parts]
    mx = max(sizes)
    buf = torch.zeros(mx, dtype=local_audio.dtype, device=local_audio.device)
    buf[: local_audio.numel()] = local_audio.reshape(-1)
    recv = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, recv, dst=dst)
    if rank != dst:
        return None
    out: List[Optional[torch.Tensor]] = [None] * len(sample_counts)
    for r, p in enumerate(parts):
        off = 0
        for i in p:
            out[i] = recv[r][off : off + int(sample_counts[i])]
            off += int(sample_counts[i])
    return out
