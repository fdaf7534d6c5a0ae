"""Helpers for multi-process CPU tests (the native socket communicator; backend="c10d" runs the
same test over a torch.distributed gloo group through the adapter)."""
import os
import socket
import sys
import tempfile
import traceback

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _entry(rank, world, port, fn, args, err_dir, backend="socket"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    if os.environ.get("PDA_TEST_DUMP_AFTER"):  # diagnosis of a stuck rank: Python stacks of all threads to stderr
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PDA_TEST_DUMP_AFTER"]), exit=False)
    try:
        from pytorch_distributed_amd.parallel import init_distributed, shutdown

        init_distributed(backend=backend, timeout_s=120)
        fn(rank, world, *args)
        shutdown()
    except BaseException:  # noqa: BLE001 - the parent reports the traceback
        with open(os.path.join(err_dir, f"rank{rank}.err"), "w") as f:
            f.write(traceback.format_exc()[-6000:])
        os._exit(1)  # do not wait in atexit handlers for peers that are stuck in a collective


def run_distributed(fn, world: int = 2, args=(), timeout: float = 240.0, backend: str = "socket"):
    """Runs ``fn(rank, world, *args)`` in ``world`` CPU processes; raises on any failure."""
    ctx = mp.get_context("spawn")
    port = free_port()
    with tempfile.TemporaryDirectory() as err_dir:
        procs = [ctx.Process(target=_entry, args=(r, world, port, fn, args, err_dir, backend)) for r in range(world)]
        for p in procs:
            p.start()
        import time

        deadline = time.time() + timeout
        while time.time() < deadline and any(p.is_alive() for p in procs):
            if any((not p.is_alive()) and p.exitcode != 0 for p in procs):
                break  # one rank failed: its peers may hang in a collective, stop waiting
            time.sleep(0.1)
        for p in procs:
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(10)
        msgs = []
        for r in range(world):
            path = os.path.join(err_dir, f"rank{r}.err")
            if os.path.exists(path):
                msgs.append(f"--- rank {r} ---\n{open(path).read()}")
        if msgs or any(p.exitcode != 0 for p in procs):
            codes = [p.exitcode for p in procs]
            raise AssertionError(f"distributed test failed (exit codes {codes}):\n" + "\n".join(msgs))
