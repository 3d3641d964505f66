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
    # ONE device -> host copy into a pinned buffer (the C-ABI's stts_load_weight takes host pointers: weight-norm folding and the Winograd /
    # Toom-Cook weight transforms run on the host, in double, once per process); the tensors are VIEWS of that buffer, nothing is copied again
    if flat.device.type == "cuda":
        pinned = torch.empty(total, dtype=torch.float32).pin_memory()
        pinned.copy_(flat, non_blocking=True)
        torch.cuda.current_stream(flat.device).synchronize()
        host = pinned.numpy()
    else:
        host = flat.numpy()
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    off = 0
    for n, s, _ in spec:
        k = int(np.prod(s))
        out[n] = host[off : off + k].reshape(s)
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


class WaveformCollector:
    """Per-batch collection of the shards' waveforms on `dst` with EXACT sizes and no per-step allocation.

    Built once for a batch layout (the per-utterance sample counts, global order): the deterministic partition gives every
    rank's payload size, rank `dst` owns one flat receive buffer laid out rank after rank, and a step is one point-to-point
    transfer per peer (``batch_isend_irecv``: over RCCL the peers write to `dst` over separate xGMI links; no ring, no
    padding to the largest shard).  ``utterance(i)`` returns a view into the receive buffer."""

    def __init__(self, sample_counts: Sequence[int], device, dst: int = 0, parts: Optional[Sequence[Sequence[int]]] = None):
        """parts: the utterance ids of every rank (ascending), when the caller already has a partition (e.g. rank r owns a fixed
        block of the global batch); default: the deterministic longest-first partition of `sample_counts`."""
        self.world, self.rank, self.dst = dist.get_world_size(), dist.get_rank(), dst
        self.counts = [int(c) for c in sample_counts]
        self.parts = [sorted(int(i) for i in p) for p in parts] if parts is not None else partition_utterances(self.counts, self.world)
        assert len(self.parts) == self.world and sorted(i for p in self.parts for i in p) == list(range(len(self.counts))), "parts must cover every utterance once"
        self.sizes = [int(sum(self.counts[i] for i in p)) for p in self.parts]
        self.offsets = [0]
        for n in self.sizes:
            self.offsets.append(self.offsets[-1] + n)
        self.local_ids = self.parts[self.rank]
        self.recv = torch.empty(self.offsets[-1], dtype=torch.float32, device=device) if self.rank == dst else None
        # gloo moves host memory only (CPU tests, single-GPU rehearsals of the N > 1 path): stage through pinned host buffers
        self._host = None
        if dist.get_backend() == "gloo" and torch.device(device).type == "cuda":
            n = self.offsets[-1] if self.rank == dst else self.sizes[self.rank]
            self._host = torch.empty(n, dtype=torch.float32).pin_memory()
        self._where = {}
        for r, p in enumerate(self.parts):
            off = self.offsets[r]
            for i in p:
                self._where[i] = (off, self.counts[i])
                off += self.counts[i]

    def collect(self, local_audio: torch.Tensor):
        """local_audio: this rank's utterances concatenated in ascending global id order (exactly sizes[rank] samples)."""
        flat = local_audio.reshape(-1)
        assert flat.numel() == self.sizes[self.rank], (flat.numel(), self.sizes[self.rank])
        if self._host is not None:
            return self._collect_via_host(flat)
        if self.rank == self.dst:
            self.recv[self.offsets[self.rank] : self.offsets[self.rank + 1]].copy_(flat)
            ops = [dist.P2POp(dist.irecv, self.recv[self.offsets[r] : self.offsets[r + 1]], r) for r in range(self.world) if r != self.dst and self.sizes[r]]
        else:
            ops = [dist.P2POp(dist.isend, flat, self.dst)] if self.sizes[self.rank] else []
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return self.recv

    def collect_async(self, local_audio: torch.Tensor, compute_stream=None, timed: bool = False):
        """collect() on the collector's OWN stream, behind what `compute_stream` (default: the current stream) has queued so far: the caller's next
        step computes while this one's waveforms travel.  Returns an event that completes when `local_audio` has been read (the caller waits for it
        before it overwrites the buffer; with two alternating audio buffers that is two steps later) - None on the CPU, where collect() is synchronous.
        timed: keep start / stop events of this transfer (collect_ms() averages them)."""
        if local_audio.device.type != "cuda":
            self.collect(local_audio)
            return None
        dev = local_audio.device
        if getattr(self, "_stream", None) is None:
            self._stream = torch.cuda.Stream(device=dev)
            self._timing = []
        cs = compute_stream if compute_stream is not None else torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(cs)
        with torch.cuda.stream(self._stream):
            self._stream.wait_event(ready)
            e0 = e1 = None
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(self._stream)
            self.collect(local_audio)
            local_audio.record_stream(self._stream)
            if timed:
                e1.record(self._stream)
                self._timing.append((e0, e1))
            done = torch.cuda.Event()
            done.record(self._stream)
        return done

    def collect_ms(self) -> Optional[float]:
        """Mean duration (ms) of the timed collect_async() transfers so far, measured on the collector's stream; None if none was timed."""
        t = getattr(self, "_timing", None)
        if not t:
            return None
        torch.cuda.synchronize()
        return float(sum(a.elapsed_time(b) for a, b in t) / len(t))

    def _collect_via_host(self, flat: torch.Tensor):
        if self.rank == self.dst:
            ops = [dist.P2POp(dist.irecv, self._host[self.offsets[r] : self.offsets[r + 1]], r) for r in range(self.world) if r != self.dst and self.sizes[r]]
        else:
            self._host.copy_(flat)
            ops = [dist.P2POp(dist.isend, self._host, self.dst)] if self.sizes[self.rank] else []
        for req in dist.batch_isend_irecv(ops) if ops else []:
            req.wait()
        if self.rank == self.dst:
            self.recv.copy_(self._host, non_blocking=True)
            self.recv[self.offsets[self.rank] : self.offsets[self.rank + 1]].copy_(flat)
        return self.recv

    def utterance(self, i: int) -> torch.Tensor:
        off, n = self._where[i]
        return self.recv[off : off + n]

    def imbalance(self) -> float:
        """max / mean of the per-rank payloads (1.0 = perfectly balanced)."""
        return max(self.sizes) / (sum(self.sizes) / self.world) if sum(self.sizes) else 1.0
