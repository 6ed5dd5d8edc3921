"""The torch-free rendezvous of the multi-GPU driver (implicit_amd/gpu/rendezvous.py): rank 0 hands a byte string to
the other ranks over TCP, addressed through the launcher's environment variables."""
import multiprocessing as mp
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port)})
    import importlib.util

    spec = importlib.util.spec_from_file_location("rendezvous", os.path.join(ROOT, "implicit_amd", "gpu", "rendezvous.py"))
    rdzv = importlib.util.module_from_spec(spec)  # loaded by path: importing the package would warn about the missing GPU
    spec.loader.exec_module(rdzv)
    assert rdzv.env_world() == (rank, world, rank)
    payload = bytes(range(128)) if rank == 0 else None
    q.put((rank, rdzv.broadcast_bytes(payload, rank, world, timeout=60.0)))


def test_broadcast_bytes_three_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 3, _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in (1, 2, 0)]  # rank 0 starts LAST: the others retry
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {r: bytes(range(128)) for r in range(world)}


def test_single_rank_needs_no_network():
    sys.path.insert(0, ROOT)
    import importlib.util

    spec = importlib.util.spec_from_file_location("rendezvous", os.path.join(ROOT, "implicit_amd", "gpu", "rendezvous.py"))
    rdzv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rdzv)
    assert rdzv.broadcast_bytes(b"abc", 0, 1) == b"abc"


def test_occupied_first_port_is_skipped():
    """MASTER_PORT + 1 may belong to somebody else: rank 0 moves on to the next candidate port, and the other ranks skip a
    listener that does not answer with the protocol's magic."""
    import threading

    ctx = mp.get_context("spawn")
    port = _free_port()
    squatter = socket.socket()
    squatter.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        squatter.bind(("", port + 1))            # the rendezvous's first choice
    except OSError:
        squatter.close()
        return                                    # the port after the free one is in use on this host: nothing to test
    squatter.listen(4)
    stop = threading.Event()

    def babble():                                 # accepts and answers garbage, like an unrelated service
        squatter.settimeout(0.2)
        while not stop.is_set():
            try:
                conn, _ = squatter.accept()
            except OSError:
                continue
            with conn:
                try:
                    conn.sendall(b"HTTP/1.0 400 nope\r\n\r\n")
                except OSError:
                    pass

    t = threading.Thread(target=babble, daemon=True)
    t.start()
    try:
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in (1, 0)]
        for p in procs:
            p.start()
        got = dict(q.get(timeout=120) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert got == {0: bytes(range(128)), 1: bytes(range(128))}
    finally:
        stop.set()
        t.join(timeout=5)
        squatter.close()
