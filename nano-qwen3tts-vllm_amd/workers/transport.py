"""Byte-message channels between the API process and the engine worker processes.

The reference uses ZeroMQ PUSH/PULL over TCP loopback (P/workers/client_bridge.py:198-246: the API process binds
four addresses, each worker connects to its command and its result address).  This module gives the same topology a
second carrier that needs nothing outside the standard library: addresses of the form `pipe://<path>` are
`multiprocessing.connection` listeners on a Unix socket - the API process listens, the worker connects - and a message
is one length-prefixed byte string, exactly the bytes workers/protocol.py serialises.  The worker entry points and
the client bridge pick the carrier from the address scheme (`tcp://...` -> ZeroMQ, `pipe://...` -> this)."""
from __future__ import annotations

import os
import tempfile
from multiprocessing.connection import Client, Connection, Listener, wait

PIPE_SCHEME = "pipe://"


def is_pipe_address(addr: str) -> bool:
    return isinstance(addr, str) and addr.startswith(PIPE_SCHEME)


def fresh_pipe_addresses(names) -> dict:
    """One Unix-socket address per name, in a private temporary directory."""
    d = tempfile.mkdtemp(prefix="qtts-pipe-")
    return {n: PIPE_SCHEME + os.path.join(d, n) for n in names}


class PipeListener:
    """The binding side of one channel (the API process).  `accept()` waits for the worker to connect."""

    def __init__(self, addr: str):
        assert is_pipe_address(addr), addr
        self.path = addr[len(PIPE_SCHEME):]
        self._listener = Listener(self.path, family="AF_UNIX")
        self.conn: Connection | None = None

    def accept(self, timeout: float = 120.0) -> Connection:
        # (Listener.accept has no timeout of its own: wait on the listening socket first)
        sock = self._listener._listener._socket
        sock.settimeout(timeout)
        try:
            self.conn = self._listener.accept()
        finally:
            sock.settimeout(None)
        return self.conn

    def close(self):
        for c in (self.conn, self._listener):
            try:
                if c is not None:
                    c.close()
            except Exception:
                pass
        try:
            os.rmdir(os.path.dirname(self.path))
        except OSError:
            pass


def connect(addr: str) -> Connection:
    """The connecting side (a worker process)."""
    assert is_pipe_address(addr), addr
    return Client(addr[len(PIPE_SCHEME):], family="AF_UNIX")


def open_worker_channels(command_connect_addr: str, result_connect_addr: str):
    """(recv, send, close) for a worker's command loop: recv() returns the next command's bytes, send(bytes) answers."""
    if is_pipe_address(command_connect_addr):
        cmd, res = connect(command_connect_addr), connect(result_connect_addr)

        def close():
            cmd.close()
            res.close()
        return cmd.recv_bytes, res.send_bytes, close
    try:
        import zmq
    except ImportError as e:
        raise ImportError("pyzmq is required for tcp:// worker addresses (pipe:// addresses need no extra package)") from e
    ctx = zmq.Context()
    pull, push = ctx.socket(zmq.PULL), ctx.socket(zmq.PUSH)
    for s, addr in ((pull, command_connect_addr), (push, result_connect_addr)):
        s.setsockopt(zmq.LINGER, 0)
        s.connect(addr)

    def close():
        pull.close()
        push.close()
        ctx.term()
    return pull.recv, push.send, close


__all__ = ["PipeListener", "connect", "fresh_pipe_addresses", "is_pipe_address", "open_worker_channels", "wait"]
