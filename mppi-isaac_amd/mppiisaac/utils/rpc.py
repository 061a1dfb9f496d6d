"""Planner RPC: the reference runs the planner and the simulated world as two processes joined by zerorpc
(reference examples/*/planner.py:43-48 `zerorpc.Server(MPPIisaacPlanner(...)).bind("tcp://0.0.0.0:4242")`,
examples/*/world.py:21-22 `zerorpc.Client().connect(...)`, then `planner.compute_action_tensor(bytes, bytes)`,
`planner.get_rollouts()`, ... with `torch.save` blobs as payload, mppiisaac/utils/transport.py:5-14).

`Server` / `Client` here have the same construction and call surface, so an example switches with
`from mppiisaac.utils import rpc as zerorpc`.  When the real `zerorpc` package is importable it is used as is
(wire-compatible with unmodified reference clients).  It is absent from this image (no pyzmq either), so the
fallback below carries the same calls over a plain TCP stream of length-prefixed msgpack frames:

    request   [msgid:int, method:str, args:list]          (bytes stay bytes: msgpack bin type)
    response  [msgid:int, error:None | [type, message, traceback], result]

One request at a time per connection (the planner is a single HIP context and not thread-safe); connections
are served one after the other in the order they arrive, like zerorpc's default single-worker server."""
import socket
import struct
import threading
import traceback
from typing import Any, Optional
from urllib.parse import urlparse

import msgpack

try:  # pragma: no cover - not present in this image
    import zerorpc as _zerorpc
except ImportError:
    _zerorpc = None

_HDR = struct.Struct("!Q")
MAX_FRAME = 1 << 31


class RemoteError(Exception):
    """an exception raised by the served object (same role as zerorpc.RemoteError)"""

    def __init__(self, name, msg, tb):
        super().__init__(f"{name}: {msg}")
        self.name, self.msg, self.traceback = name, msg, tb


class TimeoutExpired(Exception):
    pass


def _endpoint(url: str):
    u = urlparse(url)
    if u.scheme != "tcp" or u.port is None:
        raise ValueError(f"endpoint must look like tcp://host:port, got {url!r}")
    return (u.hostname or "0.0.0.0"), u.port


def _send(sock, obj):
    data = msgpack.packb(obj, use_bin_type=True)
    sock.sendall(_HDR.pack(len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        r = sock.recv_into(view[got:], n - got)
        if r == 0:
            raise ConnectionError("peer closed the connection")
        got += r
    return bytes(buf)


def _recv(sock):
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if n > MAX_FRAME:
        raise ConnectionError(f"frame of {n} bytes exceeds the limit")
    return msgpack.unpackb(_recv_exact(sock, n), raw=False, strict_map_key=False)


class _FallbackServer:
    def __init__(self, methods: Any):
        self._obj = methods
        self._sock: Optional[socket.socket] = None
        self._stop = threading.Event()
        self.endpoint = None

    def bind(self, url: str):
        host, port = _endpoint(url)
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind((host, port))
        s.listen(8)
        s.settimeout(0.2)
        self._sock = s
        self.endpoint = "tcp://%s:%d" % s.getsockname()
        return self.endpoint

    def _dispatch(self, name, args):
        if name == "_zerorpc_list":  # zerorpc's introspection call
            return sorted(k for k in dir(self._obj) if not k.startswith("_") and callable(getattr(self._obj, k)))
        if name.startswith("_"):
            raise AttributeError(f"{name!r} is not exported")
        fn = getattr(self._obj, name)
        if not callable(fn):
            raise AttributeError(f"{name!r} is not callable")
        return fn(*args)

    def _serve(self, conn):
        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        conn.settimeout(None)
        with conn:
            while not self._stop.is_set():
                try:
                    msgid, name, args = _recv(conn)
                except (ConnectionError, OSError):
                    return
                try:
                    reply = [msgid, None, self._dispatch(name, args)]
                    data_ok = True
                except Exception as e:  # the remote side sees the failure; the server keeps running
                    reply = [msgid, [type(e).__name__, str(e), traceback.format_exc()], None]
                    data_ok = False
                try:
                    _send(conn, reply)
                except TypeError as e:  # result not representable in msgpack
                    if not data_ok:
                        raise
                    _send(conn, [msgid, ["TypeError", f"result of {name} is not serialisable: {e}", ""], None])

    def run(self):
        if self._sock is None:
            raise RuntimeError("bind() first")
        try:
            while not self._stop.is_set():
                try:
                    conn, _ = self._sock.accept()
                except socket.timeout:
                    continue
                except OSError:
                    break
                self._serve(conn)
        finally:
            self.close()

    def stop(self):
        self._stop.set()

    def close(self):
        self._stop.set()
        if self._sock is not None:
            try:
                self._sock.close()
            finally:
                self._sock = None


class _FallbackClient:
    def __init__(self, connect_to: Optional[str] = None, timeout: Optional[float] = 30.0):
        self._sock: Optional[socket.socket] = None
        self._timeout = timeout
        self._msgid = 0
        self._lock = threading.Lock()
        if connect_to:
            self.connect(connect_to)

    def connect(self, url: str):
        host, port = _endpoint(url)
        s = socket.create_connection((host, port), timeout=self._timeout)
        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        s.settimeout(self._timeout)
        self._sock = s

    def __call__(self, method: str, *args):
        if self._sock is None:
            raise RuntimeError("connect() first")
        with self._lock:
            self._msgid += 1
            try:
                _send(self._sock, [self._msgid, method, list(args)])
                msgid, err, result = _recv(self._sock)
            except socket.timeout as e:
                raise TimeoutExpired(f"{method}: no reply within {self._timeout} s") from e
            if msgid != self._msgid:
                raise ConnectionError("reply does not match the request")
        if err is not None:
            raise RemoteError(*err)
        return result

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return lambda *args: self(name, *args)

    def close(self):
        if self._sock is not None:
            try:
                self._sock.close()
            finally:
                self._sock = None


if _zerorpc is not None:  # pragma: no cover
    Server, Client, BACKEND = _zerorpc.Server, _zerorpc.Client, "zerorpc"
else:
    Server, Client, BACKEND = _FallbackServer, _FallbackClient, "tcp+msgpack"
