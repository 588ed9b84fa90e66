"""Fail-fast supervision of the rank processes of a one-node data-parallel job.

The reference starts one process per GPU with ``mp.spawn`` (train.py:286) and brings the group up in
``mmidas/_dist_utils.py:43-47``; when one of those ranks dies (a device it cannot open, an RCCL init error) the others
block in their first collective until the process-group timeout.  Here the parent polls its children: on the first
non-zero exit it stops the rest, prints the failing rank's stderr tail and returns that status.

No torch import and no GPU call in this module: the parent of the ranks must not have initialised HIP.
"""
from __future__ import annotations

import os
import signal
import subprocess
import sys
import time
from typing import List, Optional, Sequence

INIT_TIMEOUT_S = 120          # process-group bring-up and the first collective (the library default would be 10 minutes)


def rank_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def _tail(path: str, n: int = 30) -> str:
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            size = f.tell()
            f.seek(max(0, size - 16384))
            return "\n".join(f.read().decode("utf-8", "replace").splitlines()[-n:])
    except OSError:
        return ""


def _stop(procs: Sequence[subprocess.Popen], grace_s: float = 5.0) -> None:
    """Terminate exactly the children this parent started (by PID), then kill what ignores the first signal."""
    for p in procs:
        if p.poll() is None:
            try:
                p.send_signal(signal.SIGTERM)
            except OSError:
                pass
    t_end = time.time() + grace_s
    for p in procs:
        while p.poll() is None and time.time() < t_end:
            time.sleep(0.05)
        if p.poll() is None:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()


def run_ranks(argv: List[str], world: int, port: int, log_dir: Optional[str] = None, poll_s: float = 0.2,
              overall_timeout_s: Optional[float] = None) -> int:
    """Start ``world`` copies of ``argv`` (rank r gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment), rank 0
    keeps this process's stdout, every rank's stderr goes to ``<log_dir>/rank<r>.err`` (and is echoed when the rank
    fails).  Returns 0 when every rank exits 0; otherwise stops the others at once and returns the first failing status."""
    import tempfile
    own_dir = None
    if not log_dir:
        own_dir = tempfile.mkdtemp(prefix="mmvae_ranks_")
        log_dir = own_dir
    os.makedirs(log_dir, exist_ok=True)
    procs, logs = [], []
    for r in range(world):
        path = os.path.join(log_dir, f"rank{r}.err")
        logs.append(path)
        err = open(path, "wb")
        out = None if r == 0 else open(os.path.join(log_dir, f"rank{r}.out"), "wb")
        procs.append(subprocess.Popen(argv, env=rank_env(r, world, port), stdout=out, stderr=err))
        err.close()
        if out is not None:
            out.close()
    t0 = time.time()
    rc = 0
    try:
        while True:
            states = [p.poll() for p in procs]
            bad = [(r, s) for r, s in enumerate(states) if s not in (None, 0)]
            if bad:
                r, s = bad[0]
                rc = abs(s) or 1
                print(f"rank {r} exited with status {s}; stopping the other ranks.  Its stderr tail ({logs[r]}):\n{_tail(logs[r])}",
                      file=sys.stderr, flush=True)
                break
            if all(s == 0 for s in states):
                break
            if overall_timeout_s is not None and time.time() - t0 > overall_timeout_s:
                rc = 124
                print(f"ranks still running after {overall_timeout_s:.0f} s; stopping them", file=sys.stderr, flush=True)
                break
            time.sleep(poll_s)
    finally:
        _stop(procs)
    if rc == 0:
        for path in logs:    # a clean run: pass the ranks' stderr through (warnings), as an unsupervised launch would
            t = _tail(path, 200)
            if t:
                print(t, file=sys.stderr, flush=True)
    return rc
