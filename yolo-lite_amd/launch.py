"""One process per GPU on one node: the launcher of the sharded path (``python bench.py --gpus N`` uses it).

The reference has no distributed code (``utils/torch_utils.py:142-159`` ends up on ``cuda:0``).  This module imports nothing
that could touch a GPU: the parent process starts N fresh children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
``torch.distributed.run`` would set them), relays one rank's stdout and every rank's stderr, and never replaces itself or a
GPU-initialised process with another program.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import List, Optional, Sequence, Tuple


def free_port() -> int:
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def launch_ranks(n: int, cmd: Sequence[str], stdout_rank: int = 0, env: Optional[dict] = None, poll_s: float = 0.05,
                 grace_s: float = 5.0) -> Tuple[int, List[str]]:
    """Start ``cmd`` N times, one rank each, and wait.  Returns ``(status, lines)``: ``lines`` = the stdout lines of rank
    ``stdout_rank`` (every other stream goes to this process's stderr, prefixed ``[rank r]``); ``status`` = 0 when every rank
    exited with 0, else the first failing rank's status (1 for a signal) -- as soon as ANY rank fails, mid-run included, the
    others are terminated (killed after ``grace_s``), so a job whose rank died never hangs in a collective and never reports
    a result."""
    port = free_port()
    procs = []
    for r in range(n):
        e = dict(os.environ if env is None else env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen(list(cmd), env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    lines: List[str] = []

    def pump(r, stream, is_out):
        for ln in stream:
            if is_out and r == stdout_rank:
                lines.append(ln)
            else:
                sys.stderr.write(f"[rank {r}] {ln}")
                sys.stderr.flush()

    th = [threading.Thread(target=pump, args=(r, s_, o), daemon=True)
          for r, p_ in enumerate(procs) for s_, o in ((p_.stdout, True), (p_.stderr, False))]
    for t in th:
        t.start()
    rc, left, t_fail = 0, set(range(n)), None
    while left:
        for r in sorted(left):
            c = procs[r].poll()
            if c is None:
                continue
            left.discard(r)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 1
                t_fail = time.monotonic()
                sys.stderr.write(f"[launch] rank {r} of {n} exited with status {c}: stopping the other ranks\n")
                for q in left:
                    procs[q].terminate()
        if t_fail is not None and left and time.monotonic() - t_fail > grace_s:
            for q in left:  # (a rank stuck in a collective may ignore SIGTERM)
                procs[q].kill()
            t_fail = time.monotonic()
        time.sleep(poll_s)
    for t in th:
        t.join(timeout=5)
    return rc, lines
