"""Start one rank process per GPU from a process that stays off the GPUs.

``python bench.py --gpus N`` (no launcher) lands here: N children of the same script with RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, rank 0's stdout relayed (it carries the one
JSON line), a non-zero exit if any rank fails.  The parent never imports torch and never makes a HIP
call -- on this platform a process that has initialised a GPU must not spawn-and-replace itself, and
there is no reason for the launcher to hold a device anyway.
"""
import json
import os
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(script, argv, n_ranks, extra_env=None, poll_s=0.2):
    """Run `python script argv...` as ranks 0..n_ranks-1; returns rank 0's last JSON line (str).
    Raises SystemExit naming the first rank that failed (the others are then terminated, by PID)."""
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(script)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout.readlines()), daemon=True)
    reader.start()            # rank 0 can never block on a full pipe
    pending = set(range(n_ranks))
    failed = None
    while pending and failed is None:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is not None:
                pending.discard(r)
                if rc != 0 and failed is None:
                    failed = (r, rc)
        if pending and failed is None:
            time.sleep(poll_s)
    if failed is not None:
        for r in pending:     # exactly the children started above
            procs[r].terminate()
        for r in pending:
            try:
                procs[r].wait(timeout=30)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    reader.join(timeout=30)
    text = [l.decode(errors="replace") for l in lines]
    if failed is not None:
        sys.stderr.write("".join(text))
        raise SystemExit("rank %d exited with status %d" % failed)
    line = None
    for l in text:
        try:
            json.loads(l)
            line = l.strip()
        except ValueError:
            sys.stderr.write(l)
    if line is None:
        raise SystemExit("rank 0 printed no JSON line")
    return line
