"""Host-side rendezvous for the one-process-per-GPU launch: barrier, tiny reductions, byte broadcast -- no PyTorch.

The data path of a sharded batch never communicates (parallel.py); what the ranks do have to exchange on the HOST is
small and rare: RCCL's 128-byte unique id (rank 0 -> everyone, once), a barrier around timed regions, and a handful of
float64 values to reduce (bench.py).  This module does exactly that over one stream socket per rank to rank 0, with the
torchrun-style environment (RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) as its only configuration:

* MASTER_ADDR local (127.0.0.1 / localhost / unset) -- the single-node case the bench contract names: an ABSTRACT
  unix-domain socket named after MASTER_PORT.  Nothing to clean up, no port to collide with: under
  `python -m torch.distributed.run` the TCP port MASTER_PORT itself is already taken by the launcher's own store.
* otherwise (several nodes): TCP to MASTER_ADDR on GU_RDZV_PORT (default MASTER_PORT + 1).  GU_RDZV=tcp forces this
  form on one node too.

Every collective is one length-prefixed message from each rank to rank 0 and one reply (star topology; a round trip is
tens of microseconds on one node).  All ranks must call the same collectives in the same order, like any such layer.
"""
import os
import socket
import struct
import time

_MAGIC = b'GURDZV2\0'


def _token():
    """16 bytes every rank of ONE launch shares (GU_RDZV_TOKEN, hex, set by whoever starts the ranks: bench.py's launcher does);
    zeros when the launcher set none.  Rank 0 turns away a hello with another token: a stray or hostile local process that
    finds the socket cannot take a rank's place."""
    try:
        raw = bytes.fromhex(os.environ.get('GU_RDZV_TOKEN', ''))
    except ValueError:
        raw = b''
    return (raw + bytes(16))[:16]


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError('rendezvous peer closed the connection')
        buf += chunk
    return bytes(buf)


def _send_msg(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)) + payload)


def _recv_msg(sock):
    (n,) = struct.unpack('<Q', _recv_exact(sock, 8))
    return _recv_exact(sock, n)


def _address():
    """(family, address) every rank derives identically from the environment."""
    addr = os.environ.get('MASTER_ADDR', '127.0.0.1')
    port = int(os.environ.get('MASTER_PORT', '29500'))
    local = addr in ('127.0.0.1', 'localhost', '::1', '')
    if local and os.environ.get('GU_RDZV', 'auto') != 'tcp' and hasattr(socket, 'AF_UNIX'):
        return socket.AF_UNIX, '\0gu-rdzv-%d' % port
    return socket.AF_INET, (addr if not local else '127.0.0.1', int(os.environ.get('GU_RDZV_PORT', port + 1)))


class Rendezvous(object):
    """rank / world from the arguments or the torchrun-style environment; a no-op for one process."""

    def __init__(self, rank=None, world=None, timeout=600.0, join_timeout=None):
        """`join_timeout`: how long to wait for every rank to JOIN (default: `timeout`); a launcher that starts and watches its
        ranks itself passes something short, so that a rank that never came up is an error within a minute or two."""
        self.rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
        self.world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
        self.peers, self.sock, self.listener = [], None, None
        if self.world <= 1:
            return
        family, address = _address()
        join_timeout = timeout if join_timeout is None else float(join_timeout)
        deadline = time.time() + join_timeout
        if self.rank == 0:
            self.listener = socket.socket(family, socket.SOCK_STREAM)
            if family == socket.AF_INET:
                self.listener.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            self.listener.bind(address)
            self.listener.listen(self.world)
            slots = [None] * self.world
            while sum(s is not None for s in slots[1:]) < self.world - 1:
                self.listener.settimeout(max(0.05, deadline - time.time()))
                try:
                    conn, _ = self.listener.accept()
                except socket.timeout:
                    raise TimeoutError('rendezvous: %d of %d ranks joined within %.0f s' % (1 + sum(s is not None for s in slots[1:]), self.world, join_timeout))
                if family == socket.AF_UNIX and hasattr(socket, 'SO_PEERCRED'):  # only this user's processes may claim a rank
                    cred = conn.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize('3i'))
                    if struct.unpack('3i', cred)[1] != os.getuid():
                        conn.close()
                        continue
                conn.settimeout(5.0)  # the hello is 16 bytes sent right after connect: a silent peer must not stall the others
                try:
                    hello = _recv_exact(conn, len(_MAGIC) + 8 + 16)
                except (ConnectionError, OSError):
                    conn.close()  # a caller that gave up (or something else that found the socket)
                    continue
                peer, peer_world = struct.unpack('<II', hello[len(_MAGIC):len(_MAGIC) + 8])
                if hello[:len(_MAGIC)] != _MAGIC or hello[len(_MAGIC) + 8:] != _token() or peer_world != self.world or not 0 < peer < self.world or slots[peer] is not None:
                    conn.close()  # not one of ours (or a leftover of another run): ignore it
                    continue
                conn.settimeout(timeout)
                self._tune(conn, family)
                slots[peer] = conn
            self.peers = slots
            for conn in self.peers[1:]:
                conn.sendall(_MAGIC)
        else:
            # connect, say hello, wait for rank 0's go-ahead (sent once every rank has joined); anything that goes wrong on the way
            # -- rank 0 not up yet, or a listener of an EARLIER rendezvous on the same name that is just being closed -- is retried
            while True:
                s = socket.socket(family, socket.SOCK_STREAM)
                try:
                    s.connect(address)
                    s.settimeout(max(1.0, deadline - time.time()))
                    self._tune(s, family)
                    s.sendall(_MAGIC + struct.pack('<II', self.rank, self.world) + _token())
                    if _recv_exact(s, len(_MAGIC)) != _MAGIC:
                        raise ConnectionError('unexpected answer')
                    break
                except (ConnectionError, FileNotFoundError, socket.timeout, OSError):
                    s.close()
                    if time.time() > deadline:
                        raise TimeoutError('rendezvous: rank 0 did not come up at %r' % (address,))
                    time.sleep(0.05)
            s.settimeout(timeout)
            self.sock = s

    @staticmethod
    def _tune(sock, family):
        if family == socket.AF_INET:
            sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    # ------------------------------------------------------------------ the one primitive
    def allgather_bytes(self, payload):
        """[world] byte strings, rank-major, identical on every rank."""
        payload = bytes(payload)
        if self.world <= 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv_msg(conn) for conn in self.peers[1:]]
            blob = b''.join(struct.pack('<Q', len(p)) + p for p in parts)
            for conn in self.peers[1:]:
                _send_msg(conn, blob)
            return parts
        _send_msg(self.sock, payload)
        blob = _recv_msg(self.sock)
        parts, off = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from('<Q', blob, off)
            parts.append(blob[off + 8:off + 8 + n])
            off += 8 + n
        return parts

    # ------------------------------------------------------------------ what callers use
    def barrier(self):
        self.allgather_bytes(b'')

    def broadcast_bytes(self, payload, src=0):
        return self.allgather_bytes(payload if self.rank == src else b'')[src]

    def gather(self, values):
        """[world][len] of every rank's list of floats."""
        mine = struct.pack('<%dd' % len(values), *[float(v) for v in values])
        return [list(struct.unpack('<%dd' % (len(p) // 8), p)) for p in self.allgather_bytes(mine)]

    def reduce(self, values, op):
        """Element-wise 'MAX' / 'MIN' / 'SUM' over ranks of a list of floats (every rank gets the result)."""
        rows = self.gather(values)
        fn = {'MAX': max, 'MIN': min, 'SUM': sum}[op]
        return [float(fn(col)) for col in zip(*rows)] if rows and rows[0] else []

    def gather_bytes(self, payload):
        return self.allgather_bytes(payload)

    def close(self):
        for conn in self.peers[1:] if self.peers else []:
            try:
                conn.close()
            except OSError:
                pass
        for s in (self.sock, self.listener):
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self.peers, self.sock, self.listener = [], None, None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


_default = None


def default():
    """The process-wide rendezvous built from the environment on first use (what parallel.py falls back to)."""
    global _default
    if _default is None:
        _default = Rendezvous()
    return _default


def broadcast_bytes(payload, src=0):
    return default().broadcast_bytes(payload, src)
