"""Starting bench.py's ranks: `python bench.py --gpus N` started plainly becomes a launcher of N child processes, before
anything in it has touched a GPU (never an exec from a process that has initialised HIP)."""
import os
import socket
import subprocess
import sys
import time

import griduniverse_amd as gua
from griduniverse_amd import _lib

from .timing import native_stdout_to_stderr


def ensure_library_is_current(engine_cls, local_rank=0):
    """A checkout whose libgu.so is missing or older than its sources: build it (one rank per node) rather than measure
    nothing.  _lib.is_stale() reads the hash from the file's bytes, so looking never maps the library."""
    if engine_cls is not gua.Engine or not _lib.is_stale():
        return
    if local_rank == 0:
        with native_stdout_to_stderr():  # (make's and hipcc's chatter belongs on stderr: stdout carries the one JSON line)
            _lib.build()
    else:
        deadline = time.time() + 900
        while _lib.is_stale() and time.time() < deadline:
            time.sleep(2)


def spawn_ranks(args, argv, script, engine_cls=None):
    """`python bench.py --gpus N` started plainly (no WORLD_SIZE in the environment): this process becomes a launcher.  It
    starts N fresh children -- one rank each, torchrun-style environment, each the leader of its own process group -- BEFORE
    anything here has touched a GPU or loaded libgu.so, relays rank 0's JSON line, and returns the worst exit code.  (Never an
    exec of a process that has initialised the GPU: the children are ordinary subprocesses and this parent never calls into HIP.)
    EVERY child is watched: the first one that dies with an error takes the others down with it at once -- the survivors would
    otherwise sit in the rendezvous until its timeout, silently, holding their GPUs -- and SIGTERM / SIGINT to the launcher
    (an outer `timeout`) are passed on to all of them."""
    import signal
    import threading

    ensure_library_is_current(engine_cls or gua.Engine)  # a subprocess `make`: no HIP call in this process
    with socket.socket() as sck:
        sck.bind(('127.0.0.1', 0))
        port = sck.getsockname()[1]
    token = os.urandom(16).hex()  # the ranks of THIS launch (griduniverse_amd/rendezvous.py turns away anyone else)
    procs = []

    def kill_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, _frame):
        kill_all(signal.SIGTERM)
        time.sleep(0.5)
        kill_all(signal.SIGKILL)
        sys.exit(128 + signum)

    previous = {sig: signal.signal(sig, on_signal) for sig in (signal.SIGTERM, signal.SIGINT)}
    out = []
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), GU_RDZV_JOIN_TIMEOUT=os.environ.get('GU_RDZV_JOIN_TIMEOUT', '120'), GU_RDZV_TOKEN=token)
            procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=env, start_new_session=True,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
        reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        failed = None
        while any(p.poll() is None for p in procs):
            failed = next(((r, p.returncode) for r, p in enumerate(procs) if p.poll() not in (None, 0)), None)
            if failed is not None:
                sys.stderr.write('bench.py: rank %d exited with code %d: stopping the other ranks\n' % failed)
                kill_all(signal.SIGTERM)
                deadline = time.time() + 5
                while time.time() < deadline and any(p.poll() is None for p in procs):
                    time.sleep(0.05)
                kill_all(signal.SIGKILL)
                break
            time.sleep(0.05)
        codes = [p.wait() for p in procs]
        reader.join(timeout=5)
    finally:
        kill_all(signal.SIGKILL)
        for sig, handler in previous.items():
            signal.signal(sig, handler)
    if out and failed is None:
        sys.stdout.write(out[0].decode('utf-8', 'replace'))
        sys.stdout.flush()
    worst = failed[1] if failed is not None else next((c for c in codes if c != 0), 0)
    if worst:
        sys.stderr.write('bench.py: rank exit codes %r\n' % (codes,))
    return worst

