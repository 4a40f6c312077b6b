"""One worker process per GPU, started by the program itself.

The reference starts its own workers (`utils.setup_envs`, utils.py:144-157: one Ray actor process per environment); here
the unit is one process per MI355X (`flingbot_amd/distributed.py`), and `launch_local_ranks` is what starts them when
nobody else did: `python bench.py --gpus 4` re-runs the same script four times with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set, exactly the environment `torch.distributed.run` would provide.

This module is standard library only and must stay that way: the parent process may not touch the GPU before (or after)
it starts the children -- on this pool replacing or forking a process that has initialised HIP takes the machine down --
so it imports neither torch nor libflingsim.  Children are fresh interpreters (subprocess, never fork / exec of self).
"""
import os
import signal
import socket
import subprocess
import sys
import time


def free_port(host="127.0.0.1"):
    s = socket.socket()
    s.bind((host, 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_local_ranks(n_ranks, script, argv, env=None, timeout=None, master_addr="127.0.0.1", master_port=None):
    """Start `n_ranks` copies of `python script argv...`, rank r with RANK = LOCAL_RANK = r, and wait for all of them.
    stdout / stderr are inherited (rank 0 prints the result line).  If a rank fails the others are terminated.
    Returns the first non-zero exit code, or 0."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    base = dict(os.environ if env is None else env)
    base["WORLD_SIZE"] = str(n_ranks)
    base["MASTER_ADDR"] = master_addr
    base["MASTER_PORT"] = str(master_port or free_port(master_addr))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this image (see README)
    procs = []
    for r in range(n_ranks):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), LOCAL_WORLD_SIZE=str(n_ranks))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=e))
    deadline = None if timeout is None else time.monotonic() + timeout
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0 or (deadline is not None and time.monotonic() > deadline):
                if rc == 0:
                    rc = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:  # exactly the PIDs started here
            if p.poll() is None:
                p.send_signal(signal.SIGTERM)
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc
