"""Rank rendezvous for the multi-GPU driver without torch: the only thing the ranks must agree on before RCCL
exists is the 128-byte communicator id, which rank 0 creates (imp_comm_unique_id) and hands to the others over
one short-lived TCP connection each.  Everything after that -- barriers, the max over ranks of the timed
region -- goes through the RCCL communicator itself (implicit_amd.gpu.Comm).

The launcher's environment is the one `python -m torch.distributed.run` / torchrun sets (the bench driver uses
it): RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR, MASTER_PORT.  MASTER_PORT itself belongs to the launcher's own
store, so the exchange listens on MASTER_PORT + 1 + IMP_RDZV_PORT_OFFSET (override the absolute port with
IMP_RDZV_PORT) or, if that is taken, on one of the 7 ports after it; peers are recognised by a magic + job-token
handshake (two jobs with overlapping candidate ranges cannot serve each other) and count once they ACK the payload.  No reference counterpart (implicit/gpu/als.cu:169 "TODO: multi-gpu support").
"""
import errno
import os
import socket
import struct
import time

_MAGIC = b"IMPRDZV1"


def env_world():
    """(rank, world_size, local_rank) from the launcher's environment (defaults: a single rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    return rank, world, int(os.environ.get("LOCAL_RANK", str(rank)))


_CANDIDATES = 8  # consecutive ports tried when the first choice is taken


def _endpoints():
    """(address, [candidate ports]).  Rank 0 listens on the first candidate it can bind; the others try the candidates in
    turn and only accept a peer that answers with the protocol's magic, so a port that belongs to somebody else (the
    launcher's own store sits on MASTER_PORT; anything may sit on MASTER_PORT + 1) is skipped, not trusted."""
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    if "IMP_RDZV_PORT" in os.environ:
        first = int(os.environ["IMP_RDZV_PORT"])
    else:
        first = int(os.environ.get("MASTER_PORT", "29500")) + 1 + int(os.environ.get("IMP_RDZV_PORT_OFFSET", "0"))
    return addr, [1024 + (first + i - 1024) % (65536 - 1024) for i in range(_CANDIDATES)]


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection early")
        buf += chunk
    return buf


def job_token(world):
    """16 bytes that identify THIS job: two jobs on one host whose candidate port ranges overlap (adjacent MASTER_PORTs)
    must not hand each other their communicator ids.  Derived from what every rank of a job shares and no other job
    does: the launcher's endpoint, its run id and the world size (IMP_RDZV_TOKEN overrides)."""
    import hashlib

    seed = os.environ.get("IMP_RDZV_TOKEN") or "|".join([
        os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "29500"),
        os.environ.get("TORCHELASTIC_RUN_ID", ""), str(world)])
    return hashlib.sha256(seed.encode()).digest()[:16]


_HELLO_TIMEOUT = 5.0  # a connection that does not say hello at once is not one of ours (port scan, health check)
_ACK = b"IMPRDZVK"


def _serve(srv, payload, world, token, deadline):
    """Rank 0: hand `payload` to each of the world - 1 peers.  One connection at a time; a stray, silent, foreign-job or
    vanished connection costs at most _HELLO_TIMEOUT and never counts; a peer counts once it has ACKed the payload."""
    served = set()
    while len(served) < world - 1:
        left = deadline - time.time()
        if left <= 0:
            raise TimeoutError(f"rendezvous: only {len(served)} of {world - 1} peers arrived")
        srv.settimeout(min(left, 30.0))
        try:
            conn, _ = srv.accept()
        except socket.timeout:
            continue
        with conn:
            try:
                conn.settimeout(_HELLO_TIMEOUT)
                hello = _recv_exact(conn, len(_MAGIC) + len(token) + 4)
                if hello[:len(_MAGIC)] != _MAGIC or hello[len(_MAGIC):len(_MAGIC) + len(token)] != token:
                    continue  # not one of ours / another job's peer: closing without the magic makes it try the next port
                peer = struct.unpack("<i", hello[len(_MAGIC) + len(token):])[0]
                if not 0 < peer < world:
                    continue
                conn.settimeout(30.0)
                conn.sendall(_MAGIC + token + struct.pack("<q", len(payload)) + payload)
                if _recv_exact(conn, len(_ACK)) == _ACK:
                    served.add(peer)
            except (ConnectionError, OSError):
                continue  # that peer retries


def broadcast_bytes(payload, rank, world, timeout=300.0):
    """Rank 0 passes `payload` (bytes), the others pass None; every rank returns rank 0's bytes."""
    if world == 1:
        return payload
    addr, ports = _endpoints()
    token = job_token(world)
    deadline = time.time() + timeout
    if rank == 0:
        srv, last = None, None
        for port in ports:
            # Every interface by default: MASTER_ADDR may be a host name that resolves to a loopback address ON the master
            # (Debian maps the host name to 127.0.1.1) while the peers on other nodes resolve it to the routable one -- bound
            # to the loopback only they would be refused until the timeout.  IMP_RDZV_BIND_ADDR=<ip> restricts the listener
            # to one interface (single-node jobs on a shared host: 127.0.0.1).
            for host in ((os.environ["IMP_RDZV_BIND_ADDR"],) if os.environ.get("IMP_RDZV_BIND_ADDR") else ("",)):
                cand = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                cand.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    cand.bind((host, port))
                    cand.listen(world)
                    srv = cand
                    break
                except OSError as e:  # taken (or not a local address): next choice
                    last = e
                    cand.close()
                    if getattr(e, "errno", None) == errno.EADDRINUSE:
                        break
            if srv is not None:
                break
        if srv is None:
            raise OSError(f"rendezvous: none of the ports {ports[0]}..{ports[-1]} could be bound: {last}")
        try:
            _serve(srv, payload, world, token, deadline)
        finally:
            srv.close()
        return payload
    last = None
    while time.time() < deadline:
        for port in ports:
            try:
                with socket.create_connection((addr, port), timeout=5.0) as conn:
                    conn.settimeout(10.0)
                    conn.sendall(_MAGIC + token + struct.pack("<i", rank))
                    if _recv_exact(conn, len(_MAGIC) + len(token)) != _MAGIC + token:
                        continue  # somebody else's service, or another job's rank 0, on this port
                    conn.settimeout(timeout)  # it is ours: from here on wait as long as the job allows
                    n = struct.unpack("<q", _recv_exact(conn, 8))[0]
                    data = _recv_exact(conn, n)
                    conn.sendall(_ACK)
                    return data
            except (ConnectionError, OSError) as e:  # nobody there (yet), or a foreign service that does not answer
                last = e
        time.sleep(0.2)
    raise TimeoutError(f"rendezvous with rank 0 at {addr}:{ports[0]}..{ports[-1]} failed: {last}")


def init_comm(gpu, rank=None, world=None, local_rank=None):
    """Sets the device, exchanges the RCCL id and returns an implicit_amd.gpu.Comm for this rank."""
    env = env_world()
    rank = env[0] if rank is None else rank
    world = env[1] if world is None else world
    local_rank = env[2] if local_rank is None else local_rank
    gpu.set_device(local_rank)
    uid = broadcast_bytes(gpu.Comm.unique_id() if rank == 0 else None, rank, world)
    return gpu.Comm(uid, world, rank)
