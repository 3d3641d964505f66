"""The N > 1 path on real hardware: bench.py's sharded workload over RCCL (one process per GPU, torch.distributed.run),
run only where at least two GPUs are visible (the round-end 8-GPU node); skipped on the one-GPU boxes.  The CPU suite covers
the same plumbing with gloo (tests/test_sharding_gloo.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_sharded_workload_over_rccl():
    n = torch.cuda.device_count()  # (counting devices does not initialise the GPU in this process)
    if n < 2:
        pytest.skip("needs >= 2 GPUs")
    n = min(n, 4)
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", "29571",
            os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = {}
    for tag, extra in (("cfg4", ["--workload", "cfg4", "--utterances", "64"]), ("cfg2", [])):
        r = subprocess.run(base + extra, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        out[tag] = json.loads(line)
    assert out["cfg4"]["n_gpus"] == n and out["cfg4"]["scaling"] == "strong" and out["cfg4"]["value"] > 0
    assert len(out["cfg4"]["config"]["frames_per_rank"]) == n and out["cfg4"]["config"]["imbalance_max_over_mean"] < 1.25
    assert out["cfg2"]["n_gpus"] == n and out["cfg2"]["scaling"] == "weak" and out["cfg2"]["config"]["global_batch"] == 8 * n
