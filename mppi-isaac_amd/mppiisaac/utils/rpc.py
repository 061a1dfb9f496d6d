"""Planner RPC: the reference runs the planner and the simulated world as two processes joined by zerorpc
(reference examples/*/planner.py:43-48 `zerorpc.Server(MPPIisaacPlanner(...)).bind("tcp://0.0.0.0:4242")`,
examples/*/world.py:21-22 `zerorpc.Client().connect(...)`, then `planner.compute_action_tensor(bytes, bytes)`,
`planner.get_rollouts()`, ... with `torch.save` blobs as payload, mppiisaac/utils/transport.py:5-14; pinned
zerorpc 0.6.3 over pyzmq 25.1 / libzmq 4.3, reference poetry.lock:2947,2122).

`Server` / `Client` here have the same construction and call surface, so an example switches with
`from mppiisaac.utils import rpc as zerorpc`.  When the real `zerorpc` package is importable it is used as is.
It is absent from this image (no pyzmq either), so this module speaks the two public protocols itself, over plain
sockets, so that an UNMODIFIED reference peer (a stock zerorpc client or server) is what sits at the other end:

ZMTP 3.0 (rfc.zeromq.org/spec/23), the framing libzmq puts on a tcp:// connection
    greeting   64 bytes: FF <8 padding> 7F | 03 00 | "NULL" padded to 20 | as-server 00 | 31 zero bytes
               (a libzmq peer announces 3.1 and sends its first 11 bytes on their own: any minor version is accepted, the
               peer falls back to the lower of the two, and partial reads are handled)
    handshake  NULL mechanism: each side sends ONE command frame  04 <size> 05 "READY" + properties
               (<1-byte name length> name <4-byte value length> value): Socket-Type = ROUTER (server) / DEALER (client),
               Identity = ""
    traffic    frames  <flags> <size: 1 byte, or 8 bytes network order when flags & 2> <body>; flags bit 0 = MORE (another
               frame of the same message follows), bit 1 = LONG, bit 2 = COMMAND.  PING commands (ZMTP 3.1 peers with
               heartbeats configured) are answered with PONG, ERROR commands raise.
zerorpc protocol v3 (zerorpc-python doc/protocol.md), what zerorpc puts into the messages
    message    [ empty delimiter frame, msgpack( [header, name, args] ) ] - what a DEALER / ROUTER pair carries
    header     {"message_id": unique id, "v": 3, "response_to": id of the request that opened the channel (replies only)}
    request    name = method, args = positional arguments (bytes stay bytes: msgpack bin type)
    reply      name = "OK", args = [result]   |   name = "ERR", args = [exception name, message, traceback]
    heartbeat  name = "_zpc_hb", args = [0], every 5 s on an open channel in both directions (a peer that hears nothing for
               two periods gives the call up: LostRemote); "_zpc_more" (stream flow control) is accepted and ignored,
               streaming replies are not produced
    builtins   _zerorpc_ping, _zerorpc_list, _zerorpc_name, _zerorpc_inspect

Conformance is to the published specifications and pinned by tests/test_rpc.py against hand-written byte sequences (the
RFC's worked greeting and READY command; a raw peer built from literal bytes in both directions).  It has NOT been run
against a live libzmq / zerorpc peer - neither package exists in this image and there is no network - and says so here.

One request at a time: the planner is a single HIP context and not thread-safe, so the server runs calls under one lock
(every connection has its own thread for framing and heartbeats)."""
import select
import socket
import struct
import threading
import time
import traceback
import uuid
from typing import Any, List, Optional
from urllib.parse import urlparse

import msgpack

try:  # pragma: no cover - not present in this image
    import zerorpc as _zerorpc
except ImportError:
    _zerorpc = None

MAX_FRAME = 1 << 28        # 256 MiB: far above the largest real payload (torch.save blobs of state / rollout tensors), far below what
                           # an unauthenticated peer may make this process allocate with a nine-byte header
FRAME_TIMEOUT = 30.0       # [s] a frame whose first byte has arrived must arrive whole within this time (half-sent frames do not pin memory)
MAX_CONNECTIONS = 32       # concurrent peers of one server (one thread each)
_CHUNK = 1 << 20
HEARTBEAT = 5.0        # zerorpc's default heartbeat period [s]

# ---------------------------------------------------------------------------------------------- ZMTP 3.0
ZMTP_SIGNATURE_PADDING = b"\x00\x00\x00\x00\x00\x00\x00\x01"   # "not significant"; libzmq's value reads as a ZMTP 1.0 length
FLAG_MORE, FLAG_LONG, FLAG_COMMAND = 1, 2, 4
# socket types a ROUTER / DEALER may talk to (RFC 28 request-reply pattern)
_COMPATIBLE = {b"ROUTER": (b"REQ", b"DEALER", b"ROUTER"), b"DEALER": (b"REP", b"DEALER", b"ROUTER")}


class ProtocolError(ConnectionError):
    """the peer does not speak ZMTP 3.x with the NULL mechanism, or broke the framing"""


def zmtp_greeting(as_server: bool = False, minor: int = 0) -> bytes:
    g = b"\xff" + ZMTP_SIGNATURE_PADDING + b"\x7f" + bytes([3, minor]) + b"NULL".ljust(20, b"\x00") + (b"\x01" if as_server else b"\x00") + b"\x00" * 31
    assert len(g) == 64
    return g


def zmtp_frame(body: bytes, more: bool = False, command: bool = False) -> bytes:
    flags = (FLAG_MORE if more else 0) | (FLAG_COMMAND if command else 0)
    if len(body) > 255:
        return bytes([flags | FLAG_LONG]) + struct.pack("!Q", len(body)) + body
    return bytes([flags, len(body)]) + body


def zmtp_command(name: bytes, data: bytes = b"") -> bytes:
    return zmtp_frame(bytes([len(name)]) + name + data, command=True)


def zmtp_ready(socket_type: bytes, identity: bytes = b"") -> bytes:
    props = b""
    for k, v in ((b"Socket-Type", socket_type), (b"Identity", identity)):
        props += bytes([len(k)]) + k + struct.pack("!I", len(v)) + v
    return zmtp_command(b"READY", props)


def parse_properties(data: bytes) -> dict:
    out, i = {}, 0
    while i < len(data):
        n = data[i]
        name = data[i + 1:i + 1 + n]
        i += 1 + n
        if i + 4 > len(data):
            raise ProtocolError("truncated metadata")
        (m,) = struct.unpack("!I", data[i:i + 4])
        i += 4
        if i + m > len(data):
            raise ProtocolError("truncated metadata value")
        out[name.lower()] = data[i:i + m]     # property names are case-insensitive
        i += m
    return out


class ZmtpConnection:
    """one established TCP connection speaking ZMTP 3.x / NULL: handshake(), then whole multipart messages"""

    def __init__(self, sock: socket.socket, socket_type: bytes):
        self.sock, self.socket_type = sock, socket_type
        self.peer_type: Optional[bytes] = None
        self.peer_minor = 0
        self._wlock = threading.Lock()
        sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)

    # -- raw
    def _recv_exact(self, n: int, deadline: Optional[float] = None) -> bytes:
        """n bytes; `deadline` (time.monotonic()): the whole read must be over by then - the wait for each piece is select() with the
        time that is left, so a peer that drips a byte every few seconds cannot hold a half-received frame for ever, and the
        socket's own time-out (shared with the sender threads' sendall) is not touched"""
        # the buffer GROWS with the bytes received (1 MiB at a time): the announced length alone allocates nothing
        buf = bytearray(min(n, _CHUNK))
        got = 0
        while got < n:
            if got == len(buf):
                buf.extend(bytes(min(n - got, _CHUNK)))
            if deadline is not None:
                left = deadline - time.monotonic()
                if left <= 0 or not select.select([self.sock], [], [], left)[0]:
                    raise ConnectionError("a frame stalled half-way")
            r = self.sock.recv_into(memoryview(buf)[got:], len(buf) - got)
            if r == 0:
                raise ConnectionError("peer closed the connection")
            got += r
        return bytes(buf)

    def _send(self, data: bytes):
        with self._wlock:
            self.sock.sendall(data)

    def _recv_frame(self):
        flags = self._recv_exact(1)[0]
        if flags & ~(FLAG_MORE | FLAG_LONG | FLAG_COMMAND):
            raise ProtocolError(f"reserved frame flag bits set: {flags:#04x}")
        # between messages a peer may stay silent for as long as the socket's time-out lets it; once a frame has started, ALL of it is
        # due within FRAME_TIMEOUT (one deadline for the frame, not a time-out per recv())
        deadline = time.monotonic() + FRAME_TIMEOUT
        try:
            n = struct.unpack("!Q", self._recv_exact(8, deadline))[0] if flags & FLAG_LONG else self._recv_exact(1, deadline)[0]
            if n > MAX_FRAME:
                raise ProtocolError(f"frame of {n} bytes exceeds the limit")
            return flags, self._recv_exact(n, deadline)
        except socket.timeout as e:
            raise ConnectionError("a frame stalled half-way") from e

    # -- handshake
    def handshake(self):
        self._send(zmtp_greeting())
        g = self._recv_exact(64)       # (a libzmq peer sends 11 bytes, then the rest once it has seen ours: partial reads are fine)
        if g[0] != 0xFF or g[9] != 0x7F:
            raise ProtocolError("not a ZMTP greeting (signature)")
        if g[10] < 3:
            raise ProtocolError(f"peer speaks ZMTP {g[10]}.x; 3.0 is required")
        self.peer_minor = g[11]
        mech = g[12:32].rstrip(b"\x00")
        if mech != b"NULL":
            raise ProtocolError(f"peer asks for security mechanism {mech!r}; only NULL is supported")
        self._send(zmtp_ready(self.socket_type))
        flags, body = self._recv_frame()
        if not flags & FLAG_COMMAND or not body:
            raise ProtocolError("expected the READY command")
        n = body[0]
        name, data = body[1:1 + n], body[1 + n:]
        if name == b"ERROR":
            raise ProtocolError("peer refused the handshake: " + data[1:1 + (data[0] if data else 0)].decode("ascii", "replace"))
        if name != b"READY":
            raise ProtocolError(f"expected READY, got command {name!r}")
        self.peer_type = parse_properties(data).get(b"socket-type", b"").upper()
        if self.peer_type not in _COMPATIBLE[self.socket_type]:
            self._send(zmtp_command(b"ERROR", bytes([19]) + b"invalid socket type"))
            raise ProtocolError(f"a {self.socket_type.decode()} socket cannot talk to a {self.peer_type.decode() or '?'} socket")

    # -- messages
    def send_message(self, frames: List[bytes]):
        self._send(b"".join(zmtp_frame(f, more=i + 1 < len(frames)) for i, f in enumerate(frames)))

    def recv_message(self) -> List[bytes]:
        frames: List[bytes] = []
        while True:
            flags, body = self._recv_frame()
            if flags & FLAG_COMMAND:
                n = body[0] if body else 0
                name, data = body[1:1 + n], body[1 + n:]
                if name == b"PING":           # ZMTP 3.1: 2-byte TTL, then up to 16 bytes of context echoed by the PONG
                    self._send(zmtp_command(b"PONG", data[2:18]))
                elif name == b"ERROR":
                    raise ProtocolError("peer sent ERROR: " + data[1:1 + (data[0] if data else 0)].decode("ascii", "replace"))
                continue                      # (PONG, SUBSCRIBE ...: nothing to do on a ROUTER / DEALER)
            frames.append(body)
            if not flags & FLAG_MORE:
                return frames

    def wait_readable(self, timeout: float) -> bool:
        """True when bytes are waiting (a time-out belongs BETWEEN messages: one that struck inside a frame would lose the
        stream position - the frames themselves are then read with the socket's own long time-out)"""
        r, _, _ = select.select([self.sock], [], [], max(timeout, 0.0))
        return bool(r)

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass


# ---------------------------------------------------------------------------------------------- zerorpc v3 events
def new_msgid() -> str:
    return str(uuid.uuid4())


def pack_event(name: str, args, msgid: Optional[str] = None, response_to=None) -> bytes:
    header = {"message_id": msgid or new_msgid(), "v": 3}
    if response_to is not None:
        header["response_to"] = response_to
    return msgpack.packb([header, name, list(args)], use_bin_type=True)


def unpack_event(payload: bytes):
    try:
        ev = msgpack.unpackb(payload, raw=False, strict_map_key=False)
    except Exception:  # noqa: BLE001 - a py2-era peer packs names as raw bytes that are not valid UTF-8 only by accident
        ev = msgpack.unpackb(payload, raw=True, strict_map_key=False)
    if not isinstance(ev, (list, tuple)) or len(ev) != 3 or not isinstance(ev[0], dict):
        raise ProtocolError("not a zerorpc event: expected [header, name, args]")
    header = {(k.decode() if isinstance(k, bytes) else k): v for k, v in ev[0].items()}
    name = ev[1].decode() if isinstance(ev[1], bytes) else ev[1]
    return header, name, (list(ev[2]) if isinstance(ev[2], (list, tuple)) else [ev[2]])


class RemoteError(Exception):
    """an exception raised by the served object (same role as zerorpc.RemoteError)"""

    def __init__(self, name, msg=None, tb=None):
        super().__init__(f"{name}: {msg}")
        self.name, self.msg, self.traceback = name, msg, tb


class TimeoutExpired(Exception):
    pass


class LostRemote(Exception):
    """no heartbeat from the peer for two periods (same role as zerorpc.LostRemote)"""


def _endpoint(url: str):
    u = urlparse(url)
    if u.scheme != "tcp" or u.port is None:
        raise ValueError(f"endpoint must look like tcp://host:port, got {url!r}")
    host = u.hostname or "0.0.0.0"
    return ("0.0.0.0" if host == "*" else host), u.port


# ---------------------------------------------------------------------------------------------- server (ROUTER)
class ZmtpServer:
    def __init__(self, methods: Any, name: Optional[str] = None, heartbeat: float = HEARTBEAT):
        self._obj = methods
        self._name = name or type(methods).__name__
        self._heartbeat = heartbeat
        self._sock: Optional[socket.socket] = None
        self._stop = threading.Event()
        self._call_lock = threading.Lock()      # the served object is not thread-safe
        self._conns: List[ZmtpConnection] = []
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

    def _methods(self):
        return sorted(k for k in dir(self._obj) if not k.startswith("_") and callable(getattr(self._obj, k)))

    def _dispatch(self, name, args):
        if name == "_zerorpc_ping":
            return ["pong", self._name]
        if name == "_zerorpc_list":
            return self._methods()
        if name == "_zerorpc_name":
            return self._name
        if name == "_zerorpc_inspect":
            return {"name": self._name, "methods": {k: {"args": [], "doc": (getattr(self._obj, k).__doc__ or "")} for k in self._methods()}}
        if name.startswith("_"):
            raise AttributeError(f"{name!r} is not exported")
        fn = getattr(self._obj, name)
        if not callable(fn):
            raise AttributeError(f"{name!r} is not callable")
        return fn(*args)

    def _serve(self, conn: ZmtpConnection):
        try:
            conn.sock.settimeout(10.0)
            conn.handshake()
            conn.sock.settimeout(None)
            while not self._stop.is_set():
                frames = conn.recv_message()
                envelope, payload = frames[:-1], frames[-1]   # (DEALER / REQ peers: an empty delimiter in front of the event)
                header, name, args = unpack_event(payload)
                if name in ("_zpc_hb", "_zpc_more") or "response_to" in header:
                    continue                                  # heartbeats of a channel that is closed on this side already
                channel = header.get("message_id")
                done = threading.Event()

                def beat():                                   # this side's heartbeat while the call runs
                    while not done.wait(self._heartbeat):
                        try:
                            conn.send_message(envelope + [pack_event("_zpc_hb", [0], response_to=channel)])
                        except OSError:
                            return
                hb = threading.Thread(target=beat, daemon=True)
                hb.start()
                try:
                    with self._call_lock:
                        result = self._dispatch(name, args)
                    reply = pack_event("OK", [result], response_to=channel)
                except Exception as e:  # noqa: BLE001 - the remote side sees the failure; the server keeps running
                    reply = pack_event("ERR", [type(e).__name__, str(e), traceback.format_exc()], response_to=channel)
                finally:
                    done.set()
                conn.send_message(envelope + [reply])
        except (ConnectionError, OSError, ValueError, msgpack.exceptions.ExtraData):
            pass                                              # the peer went away or broke the protocol: drop the connection
        finally:
            conn.close()
            if conn in self._conns:
                self._conns.remove(conn)

    def run(self):
        if self._sock is None:
            raise RuntimeError("bind() first")
        try:
            while not self._stop.is_set():
                try:
                    s, _ = self._sock.accept()
                except socket.timeout:
                    continue
                except OSError:
                    break
                if len(self._conns) >= MAX_CONNECTIONS:       # (one thread and up to MAX_FRAME of buffer per peer)
                    s.close()
                    continue
                conn = ZmtpConnection(s, b"ROUTER")
                self._conns.append(conn)
                threading.Thread(target=self._serve, args=(conn,), daemon=True).start()
        finally:
            self.close()

    def stop(self):
        self._stop.set()

    def close(self):
        self._stop.set()
        for c in self._conns:
            c.close()
        self._conns = []
        if self._sock is not None:
            try:
                self._sock.close()
            finally:
                self._sock = None


# ---------------------------------------------------------------------------------------------- client (DEALER)
class ZmtpClient:
    def __init__(self, connect_to: Optional[str] = None, timeout: Optional[float] = 30.0, heartbeat: float = HEARTBEAT):
        self._conn: Optional[ZmtpConnection] = None
        self._timeout, self._heartbeat = timeout, heartbeat
        self._lock = threading.Lock()
        if connect_to:
            self.connect(connect_to)

    def connect(self, url: str):
        host, port = _endpoint(url)
        s = socket.create_connection((host, port), timeout=self._timeout)
        conn = ZmtpConnection(s, b"DEALER")
        conn.handshake()
        s.settimeout(max(self._timeout or 0.0, 60.0))   # (within a frame; the call's own time-out is kept between messages)
        self._conn = conn

    def __call__(self, method: str, *args):
        if self._conn is None:
            raise RuntimeError("connect() first")
        conn = self._conn
        with self._lock:
            channel = new_msgid()
            conn.send_message([b"", pack_event(method, args, msgid=channel)])
            t0 = last_heard = last_sent = time.monotonic()
            while True:
                now = time.monotonic()
                if self._timeout is not None and now - t0 > self._timeout:
                    raise TimeoutExpired(f"{method}: no reply within {self._timeout} s")
                if now - last_heard > 2 * self._heartbeat and now - t0 > 2 * self._heartbeat:
                    raise LostRemote(f"{method}: no heartbeat from the server for {2 * self._heartbeat} s")
                if now - last_sent >= self._heartbeat:
                    conn.send_message([b"", pack_event("_zpc_hb", [0], response_to=channel)])
                    last_sent = now
                wait = self._heartbeat - (now - last_sent)
                if self._timeout is not None:
                    wait = min(wait, self._timeout - (now - t0))
                if not conn.wait_readable(max(wait, 0.01)):
                    continue
                frames = conn.recv_message()
                header, name, eargs = unpack_event(frames[-1])
                if header.get("response_to") != channel:
                    continue                                  # a late heartbeat of an earlier call
                last_heard = time.monotonic()
                if name == "_zpc_hb" or name == "_zpc_more":
                    continue
                if name == "OK":
                    return eargs[0] if eargs else None
                if name == "ERR":
                    raise RemoteError(*(eargs + [None] * 3)[:3])
                raise RemoteError("ProtocolError", f"unsupported reply event {name!r} (streams are not supported)", "")

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return lambda *args: self(name, *args)

    def close(self):
        if self._conn is not None:
            self._conn.close()
            self._conn = None


if _zerorpc is not None:  # pragma: no cover
    Server, Client, BACKEND = _zerorpc.Server, _zerorpc.Client, "zerorpc"
else:
    Server, Client, BACKEND = ZmtpServer, ZmtpClient, "zmtp3+zerorpc3 (in-tree)"
