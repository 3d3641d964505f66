"""One process per GPU, started by a parent that never touches the GPU.

`python bench.py --gpus N` (no torchrun) lands here: the parent starts N copies of the script as CHILD processes with the
torch.distributed environment contract (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, rendezvous on
127.0.0.1), relays rank 0's stdout, sends the other ranks' stdout to stderr and exits with the children's status.  Nothing here
imports torch or calls HIP: a process that has initialised the GPU must never start (exec) another program on this pool, and
a GPU-free parent may.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence, Tuple


def free_port() -> int:
    """A port that was free a moment ago (the socket is closed before the ranks bind it: launch_ranks retries with another port when the
    rendezvous finds it taken)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    return env


def launch_ranks(script: str, argv: Sequence[str], world: int, timeout: Optional[float] = None, out=None, err=None, attempts: int = 3) -> Tuple[int, List[str]]:
    """Run `script argv` as `world` rank processes.  Returns (exit status, rank 0's stdout lines).  The status is 0 only if every
    rank exited 0; when one rank fails the others are terminated (exact PIDs) and the first failing status is returned.
    A run that dies within its first seconds with the rendezvous port taken (EADDRINUSE: somebody bound it between free_port() and the
    ranks' bind) is started again on another port, up to `attempts` times."""
    err = err or sys.stderr
    for k in range(attempts):
        captured: List[str] = []

        class Tee:
            def write(self, t):
                captured.append(t)
                return err.write(t)

            def flush(self):
                err.flush()

        t0 = time.monotonic()
        status, lines = _launch_once(script, argv, world, timeout, out, Tee())
        in_use = any("EADDRINUSE" in t or "Address already in use" in t or "address already in use" in t for t in captured)
        if status == 0 or not in_use or lines or time.monotonic() - t0 > 60 or k + 1 == attempts:
            return status, lines
        err.write(f"[launcher] rendezvous port was taken: starting the ranks again on another port (attempt {k + 2} of {attempts})\n")
    return status, lines


def _launch_once(script: str, argv: Sequence[str], world: int, timeout: Optional[float], out, err) -> Tuple[int, List[str]]:
    out = out or sys.stdout
    err = err or sys.stderr
    port = free_port()
    procs: List[subprocess.Popen] = []
    lines: List[str] = []

    def relay(p: subprocess.Popen, sink, keep: Optional[List[str]], tag: str):
        for raw in iter(p.stdout.readline, b""):
            line = raw.decode("utf-8", "replace")
            # rank 0's stdout carries the result (JSON lines); library chatter on stdout (e.g. gloo's connection notes) goes to stderr
            if keep is not None and line.lstrip().startswith("{"):
                keep.append(line.rstrip("\n"))
                sink.write(line)
                sink.flush()
            else:
                err.write((tag or "[rank 0] ") + line)
                err.flush()

    threads = []
    for r in range(world):
        p = subprocess.Popen([sys.executable, script, *argv], env=rank_env(r, world, port), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        te = threading.Thread(target=lambda p=p: [err.write(raw.decode("utf-8", "replace")) for raw in iter(p.stderr.readline, b"")], daemon=True)
        te.start()
        threads.append(te)
        procs.append(p)
        t = threading.Thread(target=relay, args=(p, out if r == 0 else err, lines if r == 0 else None, "" if r == 0 else f"[rank {r}] "), daemon=True)
        t.start()
        threads.append(t)
    t0 = time.monotonic()
    status = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                status = bad[0] if bad[0] > 0 else 128 - bad[0]  # a signal's negative code becomes 128 + signal
                break
            if all(c == 0 for c in codes):
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                err.write(f"[launcher] ranks still running after {timeout:.0f} s: terminating them\n")
                status = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:  # only the processes started above, by handle
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for t in threads:
            t.join(timeout=5)
    return status, lines
