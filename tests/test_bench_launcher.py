"""`python bench.py --gpus N` as the driver types it (no torchrun): the GPU-free parent starts N ranks as child processes,
relays rank 0's one JSON line and exits with their status.  Runs on the CPU: the ranks rendezvous over gloo and run the sharded
workload's partition + exact-size waveform collection on fabricated data (`--plumbing-check`; no GPU, no compute, no metric)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(*extra, env=None):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)


@pytest.mark.timeout(300)
def test_parent_starts_two_ranks_and_relays_one_json_line():
    r = _run("--gpus", "2", "--plumbing-check")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["ok"] is True
    assert len(out["frames_per_rank"]) == 2 and out["imbalance_max_over_mean"] < 1.1


@pytest.mark.timeout(400)
def test_parent_starts_eight_ranks():
    """The driver's N = 8 command line on the CPU: eight gloo ranks, one JSON line, every utterance of the fabricated batch collected on rank 0."""
    r = _run("--gpus", "8", "--plumbing-check")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks"] == 8 and out["ok"] is True and len(out["frames_per_rank"]) == 8
    assert all(f > 0 for f in out["frames_per_rank"]) and len(out["cpus_per_rank"]) == 8


def test_rank_cpu_slices_are_disjoint():
    sys.path.insert(0, ROOT)
    import bench

    have = sorted(os.sched_getaffinity(0))
    got = []
    try:
        for r in range(2):
            got.append(bench.pin_rank_cpus(r, 2))
            os.sched_setaffinity(0, have)  # (every rank is its own process: each slices the full set)
    finally:
        os.sched_setaffinity(0, have)
    if len(have) >= 4:
        assert got[0] and got[1] and not set(got[0]) & set(got[1]) and set(got[0]) | set(got[1]) <= set(have)


@pytest.mark.timeout(300)
def test_a_failing_rank_fails_the_parent():
    r = _run("--gpus", "2", "--plumbing-check", "--fail-rank", "1")
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.strip().startswith("{")]


def test_world_size_mismatch_is_a_usage_error():
    r = _run("--gpus", "2", "--plumbing-check", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and b"--gpus 2 but WORLD_SIZE is 1" in r.stderr


def test_launcher_environment_contract():
    from stylish_tts_amd.launcher import rank_env

    e = rank_env(3, 8, 29512, base={})
    assert e["RANK"] == "3" and e["LOCAL_RANK"] == "3" and e["WORLD_SIZE"] == "8" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29512"
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
