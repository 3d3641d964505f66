"""N > 1 plumbing on CPU: world_size-2 gloo processes exercise the weight broadcast, the deterministic utterance
partition and the waveform gather (the same code bench.py / the multi-GPU driver uses over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stylish_tts_amd import params, sharding


def test_partition_is_balanced_and_complete():
    rng = np.random.default_rng(0)
    frames = (rng.uniform(0.25, 10.0, 256) * 80).round().astype(int)  # cfg4: 0.25-10 s utterances
    parts = sharding.partition_utterances(frames, 8)
    flat = sorted(i for p in parts for i in p)
    assert flat == list(range(256))
    loads = [int(frames[p].sum()) for p in parts]
    assert max(loads) - min(loads) <= frames.max()
    assert max(loads) / (sum(loads) / 8) < 1.02
    assert sharding.partition_utterances([5], 4) == [[0], [], [], []]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        spec = [("a.weight", (4, 3, 2), "w"), ("a.bias", (4,), "b"), ("g", (1, 1, 5), "grn")]
        sd = params.synth_state_dict(spec, 0, prefix="t.") if rank == 0 else None
        got = sharding.broadcast_state_dict(sd, spec, torch.device("cpu"))
        want = params.synth_state_dict(spec, 0, prefix="t.")
        ok_b = all(np.array_equal(got[k], want[k]) for k in want)
        counts = [300, 75, 150, 225, 75]
        parts = sharding.partition_utterances(counts, world)
        mine = parts[rank]
        local = torch.cat([torch.full((counts[i],), float(i)) for i in mine]) if mine else torch.zeros(0)
        out = sharding.gather_waveforms(local, mine, counts, dst=0)
        ok_g = True
        if rank == 0:
            ok_g = all(o.numel() == c and bool((o == float(i)).all()) for i, (o, c) in enumerate(zip(out, counts)))
        # the pre-sized exact-size collector (what bench.py --workload cfg4 / cfg5 uses per step): two steps, same buffers
        col = sharding.WaveformCollector(counts, torch.device("cpu"), dst=0)
        assert col.local_ids == mine
        for step in range(2):
            col.collect(local + float(step))
            if rank == 0:
                ok_g = ok_g and all(col.utterance(i).numel() == c and bool((col.utterance(i) == float(i + step)).all()) for i, c in enumerate(counts))
        ok_g = ok_g and col.imbalance() < 1.2 and col.recv is (col.recv if rank == 0 else None)
        q.put((rank, ok_b, ok_g))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_broadcast_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    assert all(b and g for _, b, g in res), res
