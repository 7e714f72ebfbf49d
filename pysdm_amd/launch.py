"""One process per GPU: starting the ranks of a multi-GPU run from a plain `python script.py`.

`torch.distributed.run` is started as a CHILD process and this process only waits for it: the
parent must not have touched the GPU (a process that has initialised HIP must never be replaced or
forked into ranks), so call `spawn_ranks` before anything imports a device runtime.  Rendezvous on
127.0.0.1 with a free port; the children find RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
environment, exactly as when the script is started through torchrun by hand.
"""
import os
import socket
import subprocess
import sys


def launched_by_torchrun(environ=None):
    """True inside a rank started by torch.distributed.run (or any launcher that sets the usual
    rendezvous variables)"""
    environ = os.environ if environ is None else environ
    return "RANK" in environ and "WORLD_SIZE" in environ


def free_port():
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def spawn_ranks(script, argv, n_ranks, *, env=None, stdout=None, timeout=None):
    """runs `script argv...` as `n_ranks` ranks of one node and returns the launcher's exit code
    (0 only if every rank exited with 0).  The ranks inherit stdout / stderr unless `stdout` is
    given."""
    if n_ranks < 1:
        raise ValueError("n_ranks")
    environ = dict(os.environ if env is None else env)
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK"):
        environ.pop(name, None)
    environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL needs dmabuf IPC on this host
    environ.setdefault("OMP_NUM_THREADS", "1")
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), script, *argv]
    done = subprocess.run(command, env=environ, stdout=stdout, timeout=timeout, check=False)
    return done.returncode
