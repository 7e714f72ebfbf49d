"""One process per GPU: starting the ranks of a multi-GPU run from a plain `python script.py`.

`torch.distributed.run` is started as a CHILD process and this process only waits for it: the
parent must not have touched the GPU (a process that has initialised HIP must never be replaced or
forked into ranks), so call `spawn_ranks` before anything imports a device runtime.  Rendezvous on
127.0.0.1 with a free port; the children find RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
environment, exactly as when the script is started through torchrun by hand.
"""
import os
import signal
import socket
import subprocess
import sys


def launched_by_torchrun(environ=None):
    """True inside a rank started by torch.distributed.run (or any launcher that sets the usual
    rendezvous variables)"""
    environ = os.environ if environ is None else environ
    return "RANK" in environ and "WORLD_SIZE" in environ


def free_port():
    """a port that was free a moment ago (for a process group of ONE rank, where nobody else
    races for it between this call and the bind; `spawn_ranks` lets the launcher pick its own)"""
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
    # --standalone: the launcher binds the rendezvous port itself (a port picked here could be
    # taken by someone else before torch.distributed.run binds it); 127.0.0.1: the container's
    # host name may not resolve
    command = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={n_ranks}", "--standalone", "--local-addr", "127.0.0.1",
               script, *argv]
    # a session of its own: on a timeout the WHOLE group goes - the launcher and the ranks it
    # started, which would otherwise keep the GPU
    with subprocess.Popen(command, env=environ, stdout=stdout, start_new_session=True) as child:
        try:
            return child.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            child.wait()
            raise
