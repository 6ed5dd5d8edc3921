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


def _worker_job(rank, world, master_port, rdzv_port, payload, q, tag):
    """Like _worker, with the rendezvous port pinned so that two jobs can be made to collide on purpose."""
    sys.path.insert(0, ROOT)
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(master_port), "IMP_RDZV_PORT": str(rdzv_port)})
    import importlib.util

    spec = importlib.util.spec_from_file_location("rendezvous", os.path.join(ROOT, "implicit_amd", "gpu", "rendezvous.py"))
    rdzv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rdzv)
    q.put((tag, rank, rdzv.broadcast_bytes(payload if rank == 0 else None, rank, world, timeout=60.0)))


def test_two_jobs_with_overlapping_ports_do_not_cross():
    """Job A's rank 0 listens on the very port job B's peers try first (adjacent MASTER_PORTs make the candidate ranges
    overlap): the job token in the handshake makes B's peer walk on to B's own rank 0 instead of taking A's id."""
    ctx = mp.get_context("spawn")
    base = _free_port()
    q = ctx.Queue()
    a0 = ctx.Process(target=_worker_job, args=(0, 2, base, base + 1, b"A" * 128, q, "A"))
    a0.start()                                    # A's rank 0 owns base + 1 ...
    import time

    time.sleep(1.0)
    procs = [ctx.Process(target=_worker_job, args=(1, 2, base + 7, base + 1, None, q, "B")),   # ... B's peer tries it first
             ctx.Process(target=_worker_job, args=(0, 2, base + 7, base + 1, b"B" * 128, q, "B")),  # B's rank 0: next free port
             ctx.Process(target=_worker_job, args=(1, 2, base, base + 1, None, q, "A"))]
    for p in procs:
        p.start()
    got = {}
    for _ in range(4):
        tag, rank, data = q.get(timeout=120)
        got[(tag, rank)] = data
    for p in [a0] + procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {("A", 0): b"A" * 128, ("A", 1): b"A" * 128, ("B", 0): b"B" * 128, ("B", 1): b"B" * 128}


def test_silent_stray_connection_does_not_stall_the_peers():
    """A connection that never says hello (health check, port scan) is dropped after a few seconds and is not counted."""
    import time

    ctx = mp.get_context("spawn")
    port = _free_port()
    q = ctx.Queue()
    r0 = ctx.Process(target=_worker, args=(0, 2, port, q))
    r0.start()
    stray = None
    for _ in range(100):                          # wait for rank 0's listener, then sit on a connection without a word
        try:
            stray = socket.create_connection(("127.0.0.1", port + 1), timeout=1.0)
            break
        except OSError:
            time.sleep(0.1)
    assert stray is not None
    r1 = ctx.Process(target=_worker, args=(1, 2, port, q))
    r1.start()
    t0 = time.time()
    got = dict(q.get(timeout=120) for _ in range(2))
    stray.close()
    for p in (r0, r1):
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {0: bytes(range(128)), 1: bytes(range(128))} and time.time() - t0 < 40
