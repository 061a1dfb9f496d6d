"""Planner RPC: the reference's two-process workflow - planner served, world as client (reference
examples/panda/planner.py:43-48, world.py:21-39) - over the in-tree implementation of the two public protocols a stock zerorpc
peer speaks: ZMTP 3.0 framing (NULL mechanism, ROUTER / DEALER) and zerorpc v3 events (mppiisaac/utils/rpc.py).  zerorpc and
pyzmq are absent from the image: conformance is pinned here by HAND-WRITTEN byte sequences - the worked greeting and READY
command of the ZMTP RFC (rfc.zeromq.org/spec/23), and raw peers assembled from literal bytes in both directions - plus a
self-loop.  Not pinned against a live libzmq peer (there is none to run)."""
import socket
import struct
import threading
import time

import msgpack
import numpy as np
import pytest
import torch

from mppiisaac.utils import rpc
from mppiisaac.utils.transport import bytes_to_torch, torch_to_bytes

# ---- literal protocol bytes, written out by hand from the specifications (NOT produced by rpc.py) ---------------------------
# RFC 23/ZMTP "worked example": signature FF + 8 octets of padding + 7F, version 3.0, mechanism "NULL" in 20 octets, as-server
# 0, 31 octets of filler = 64 octets
RFC_GREETING = bytes.fromhex("ff" + "0000000000000001" + "7f" + "0300" + "4e554c4c" + "00" * 16 + "00" + "00" * 31)
# what libzmq 4.3 (pyzmq 25.1, reference poetry.lock:2122) puts on the wire: ZMTP 3.1
LIBZMQ_GREETING = RFC_GREETING[:11] + b"\x01" + RFC_GREETING[12:]
# RFC 23: READY of a DEALER with an empty identity - command frame (flags 04), body size 41: 05 "READY", 0B "Socket-Type"
# 00000006 "DEALER", 08 "Identity" 00000000
RFC_READY_DEALER = bytes([0x04, 41, 5]) + b"READY" + bytes([11]) + b"Socket-Type" + bytes([0, 0, 0, 6]) + b"DEALER" + bytes([8]) + b"Identity" + bytes([0, 0, 0, 0])
READY_ROUTER = bytes([0x04, 41, 5]) + b"READY" + bytes([11]) + b"Socket-Type" + bytes([0, 0, 0, 6]) + b"ROUTER" + bytes([8]) + b"Identity" + bytes([0, 0, 0, 0])


class FakePlanner:
    """the call surface world.py uses (reference examples/panda/world.py:36-48)"""

    def __init__(self):
        self.calls = 0

    def compute_action_tensor(self, dof_bytes, root_bytes):
        dof, root = bytes_to_torch(dof_bytes), bytes_to_torch(root_bytes)
        self.calls += 1
        return torch_to_bytes(dof[0, 0::2] + root[0, 0, 0])

    def get_rollouts(self):
        return torch_to_bytes(torch.arange(30 * 64 * 3, dtype=torch.float32).reshape(30, 64, 3))

    def big(self, n):
        return b"\x01" * n

    def fails(self):
        raise ValueError("no such actor")

    def add_to_env(self, cfgs):
        return len(cfgs)


def serve(obj, **kw):
    srv = rpc.Server(obj, **kw)
    url = srv.bind("tcp://127.0.0.1:0") if rpc.BACKEND != "zerorpc" else None
    t = threading.Thread(target=srv.run, daemon=True)
    t.start()
    return srv, url, t


@pytest.mark.skipif(rpc.BACKEND == "zerorpc", reason="real zerorpc present: its own tests apply")
def test_planner_calls_round_trip_over_zmtp():
    srv, url, t = serve(FakePlanner())
    try:
        c = rpc.Client()
        c.connect(url)
        dof = torch.tensor([[0.1, 0.0, 0.2, 0.0, 0.3, 0.0]])
        root = torch.zeros((1, 2, 13))
        root[0, 0, 0] = 1.0
        out = bytes_to_torch(c.compute_action_tensor(torch_to_bytes(dof), torch_to_bytes(root)))
        np.testing.assert_allclose(out.numpy(), [1.1, 1.2, 1.3], rtol=1e-6)
        assert bytes_to_torch(c.get_rollouts()).shape == (30, 64, 3)
        assert c.add_to_env([{"type": "sphere", "name": "s0", "size": [0.1]}]) == 1     # plain python structures pass too
        assert len(c.big(8 << 20)) == 8 << 20                                            # multi-megabyte frames
        with pytest.raises(rpc.RemoteError, match="no such actor"):
            c.fails()
        with pytest.raises(rpc.RemoteError):
            c("_private")
        with pytest.raises(rpc.RemoteError):
            c.calls()                                                                    # attribute, not a method
        assert "compute_action_tensor" in c("_zerorpc_list")
        # the server survived the failures and serves a second client after the first one leaves
        c.close()
        c2 = rpc.Client(url)
        assert bytes_to_torch(c2.get_rollouts()).shape == (30, 64, 3)
        c2.close()
    finally:
        srv.close()
        t.join(timeout=2)
    assert not t.is_alive()


def test_wire_bytes_equal_the_rfc_examples():
    """what rpc.py emits IS the RFC's worked example, octet for octet"""
    assert len(RFC_GREETING) == 64
    assert rpc.zmtp_greeting() == RFC_GREETING
    assert rpc.zmtp_ready(b"DEALER") == RFC_READY_DEALER
    assert rpc.zmtp_ready(b"ROUTER") == READY_ROUTER
    assert rpc.zmtp_frame(b"") == b"\x00\x00" and rpc.zmtp_frame(b"", more=True) == b"\x01\x00"
    assert rpc.zmtp_frame(b"abc", more=True) == b"\x01\x03abc"
    big = b"x" * 256                                        # 256 octets and more: LONG flag, 8-octet size in network order
    assert rpc.zmtp_frame(big) == b"\x02" + struct.pack("!Q", 256) + big
    assert rpc.zmtp_frame(b"x" * 255)[:2] == b"\x00\xff"
    assert rpc.parse_properties(RFC_READY_DEALER[8:]) == {b"socket-type": b"DEALER", b"identity": b""}
    # zerorpc v3 event: msgpack array [header map, name, args]; bytes payloads as msgpack bin
    ev = msgpack.unpackb(rpc.pack_event("compute_action_tensor", [b"\x00\x01", b"\x02"], msgid="abc"), raw=False)
    assert ev == [{"message_id": "abc", "v": 3}, "compute_action_tensor", [b"\x00\x01", b"\x02"]]
    ev = msgpack.unpackb(rpc.pack_event("OK", [7], msgid="r", response_to="abc"), raw=False)
    assert ev == [{"message_id": "r", "v": 3, "response_to": "abc"}, "OK", [7]]


def _read_exact(s, n):
    out = b""
    while len(out) < n:
        chunk = s.recv(n - len(out))
        assert chunk, "peer closed"
        out += chunk
    return out


def _read_frame(s):
    flags, = _read_exact(s, 1)
    n = struct.unpack("!Q", _read_exact(s, 8))[0] if flags & 2 else _read_exact(s, 1)[0]
    return flags, _read_exact(s, n)


@pytest.mark.skipif(rpc.BACKEND == "zerorpc", reason="real zerorpc present")
def test_raw_dealer_built_from_literal_bytes_is_served():
    """a stock zerorpc CLIENT as bytes: the libzmq 4.3 greeting sent the way libzmq sends it (11 octets, then the rest), the RFC's
    DEALER READY, then [empty delimiter, msgpack event] - against rpc.Server.  The reply must be a well-formed ROUTER answer."""
    srv, url, t = serve(FakePlanner())
    try:
        host, port = url[len("tcp://"):].rsplit(":", 1)
        s = socket.create_connection((host, int(port)), timeout=5)
        s.sendall(LIBZMQ_GREETING[:11])                     # signature + major version first, as libzmq does
        got = _read_exact(s, 64)
        assert got[0] == 0xFF and got[9] == 0x7F and got[10] == 3 and got[12:16] == b"NULL" and got[32] == 0
        s.sendall(LIBZMQ_GREETING[11:] + RFC_READY_DEALER)
        flags, body = _read_frame(s)
        assert flags == 0x04 and body == READY_ROUTER[2:]   # the server's READY: Socket-Type ROUTER
        # request: header {message_id, v: 3}, name, args - packed by hand with msgpack, bytes as bin
        req = msgpack.packb([{"message_id": "11111111-2222-3333-4444-555555555555", "v": 3}, "big", [300]], use_bin_type=True)
        s.sendall(b"\x01\x00" + bytes([0x00, len(req)]) + req)        # MORE + empty delimiter, then the final frame
        f0, d0 = _read_frame(s)
        f1, d1 = _read_frame(s)
        assert (f0, d0) == (0x01, b"") and f1 in (0x00, 0x02)          # delimiter with MORE, then ONE final frame (long: > 255)
        header, name, args = msgpack.unpackb(d1, raw=False)
        assert name == "OK" and args == [b"\x01" * 300]
        assert header["v"] == 3 and header["response_to"] == "11111111-2222-3333-4444-555555555555" and header["message_id"]
        # an exception of the served object: ERR with [name, message, traceback]
        req = msgpack.packb([{"message_id": "m2", "v": 3}, "fails", []], use_bin_type=True)
        s.sendall(b"\x01\x00" + bytes([0x00, len(req)]) + req)
        _read_frame(s)
        header, name, args = msgpack.unpackb(_read_frame(s)[1], raw=False)
        assert name == "ERR" and args[0] == "ValueError" and "no such actor" in args[1] and header["response_to"] == "m2"
        # a heartbeat of the client between calls is swallowed; a ZMTP 3.1 PING command is answered with PONG + its context
        hb = msgpack.packb([{"message_id": "m3", "v": 3, "response_to": "m2"}, "_zpc_hb", [0]], use_bin_type=True)
        s.sendall(b"\x01\x00" + bytes([0x00, len(hb)]) + hb)
        s.sendall(bytes([0x04, 9, 4]) + b"PING" + b"\x00\x0a" + b"ab")
        flags, body = _read_frame(s)
        assert flags == 0x04 and body == bytes([4]) + b"PONG" + b"ab"
        req = msgpack.packb([{"message_id": "m4", "v": 3}, "_zerorpc_ping", []], use_bin_type=True)
        s.sendall(b"\x01\x00" + bytes([0x00, len(req)]) + req)
        _read_frame(s)
        header, name, args = msgpack.unpackb(_read_frame(s)[1], raw=False)
        assert name == "OK" and args == [["pong", "FakePlanner"]]
        s.close()
    finally:
        srv.close()
        t.join(timeout=2)


@pytest.mark.skipif(rpc.BACKEND == "zerorpc", reason="real zerorpc present")
def test_client_talks_to_a_raw_router_built_from_literal_bytes():
    """a stock zerorpc SERVER as bytes (ROUTER side: ZMTP 3.1 greeting, READY, a heartbeat on the channel, then OK) against
    rpc.Client: the request must arrive as [delimiter, event] with a v3 header, and the client must take the reply"""
    lst = socket.socket()
    lst.bind(("127.0.0.1", 0))
    lst.listen(1)
    seen = {}

    def router():
        s, _ = lst.accept()
        s.settimeout(5)
        s.sendall(LIBZMQ_GREETING[:11])
        seen["greeting"] = _read_exact(s, 64)
        s.sendall(LIBZMQ_GREETING[11:] + READY_ROUTER)
        seen["ready"] = _read_frame(s)
        seen["delim"] = _read_frame(s)
        flags, payload = _read_frame(s)
        header, name, args = msgpack.unpackb(payload, raw=False)
        seen["event"] = (flags, header, name, args)
        hb = msgpack.packb([{"message_id": "h1", "v": 3, "response_to": header["message_id"]}, "_zpc_hb", [0]], use_bin_type=True)
        ok = msgpack.packb([{"message_id": "r1", "v": 3, "response_to": header["message_id"]}, "OK", [[1, 2, 3]]], use_bin_type=True)
        for p in (hb, ok):
            s.sendall(b"\x01\x00" + bytes([0x00, len(p)]) + p)
        time.sleep(0.2)
        s.close()
    th = threading.Thread(target=router, daemon=True)
    th.start()
    c = rpc.Client("tcp://127.0.0.1:%d" % lst.getsockname()[1], timeout=5)
    assert c.compute(b"\x07" * 4, 2.5) == [1, 2, 3]
    c.close()
    th.join(timeout=5)
    lst.close()
    assert seen["greeting"] == RFC_GREETING and seen["ready"] == (0x04, RFC_READY_DEALER[2:])
    assert seen["delim"] == (0x01, b"")
    flags, header, name, args = seen["event"]
    assert flags == 0x00 and name == "compute" and args == [b"\x07" * 4, 2.5]
    assert header["v"] == 3 and isinstance(header["message_id"], str) and "response_to" not in header


@pytest.mark.skipif(rpc.BACKEND == "zerorpc", reason="real zerorpc present")
def test_heartbeats_keep_a_long_call_alive_and_a_silent_server_is_given_up():
    class Slow:
        def slow(self, seconds):
            time.sleep(seconds)
            return "done"
    srv, url, t = serve(Slow(), heartbeat=0.1)
    try:
        c = rpc.Client(url, timeout=5, heartbeat=0.1)
        assert c.slow(0.6) == "done"                        # six heartbeat periods: the server's _zpc_hb keep the channel alive
        c.close()
    finally:
        srv.close()
        t.join(timeout=2)
    # a ROUTER that shakes hands and then never answers: LostRemote after two periods, long before the call's time-out
    lst = socket.socket()
    lst.bind(("127.0.0.1", 0))
    lst.listen(1)

    def mute():
        s, _ = lst.accept()
        s.sendall(LIBZMQ_GREETING + READY_ROUTER)
        time.sleep(1.5)
        s.close()
    th = threading.Thread(target=mute, daemon=True)
    th.start()
    c = rpc.Client("tcp://127.0.0.1:%d" % lst.getsockname()[1], timeout=10, heartbeat=0.2)
    t0 = time.monotonic()
    with pytest.raises(rpc.LostRemote):
        c.anything()
    assert time.monotonic() - t0 < 1.4
    c.close()
    th.join(timeout=3)
    lst.close()


def test_handshake_refusals():
    """wrong signature, ZMTP 2, a security mechanism, an incompatible socket type: ProtocolError, nothing hangs"""
    if rpc.BACKEND == "zerorpc":
        pytest.skip("real zerorpc")
    bad = [b"GET / HTTP/1.1\r\n" + b"\x00" * 48,
           RFC_GREETING[:10] + b"\x01" + RFC_GREETING[11:],                       # ZMTP 1/2 revision byte
           RFC_GREETING[:12] + b"PLAIN".ljust(20, b"\x00") + RFC_GREETING[32:],    # PLAIN mechanism
           RFC_GREETING + bytes([0x04, 38, 5]) + b"READY" + bytes([11]) + b"Socket-Type" + bytes([0, 0, 0, 3]) + b"PUB" + bytes([8]) + b"Identity" + bytes([0, 0, 0, 0])]
    for blob in bad:
        lst = socket.socket()
        lst.bind(("127.0.0.1", 0))
        lst.listen(1)

        def peer():
            s, _ = lst.accept()
            s.sendall(blob)
            time.sleep(0.3)
            s.close()
        th = threading.Thread(target=peer, daemon=True)
        th.start()
        with pytest.raises(ConnectionError):
            rpc.Client("tcp://127.0.0.1:%d" % lst.getsockname()[1], timeout=3)
        th.join(timeout=3)
        lst.close()


def test_frames_are_bounded_and_stalled_frames_are_dropped(monkeypatch):
    """an unauthenticated peer cannot pin memory: a long frame beyond MAX_FRAME is refused from its nine-byte header, the buffer
    of an accepted frame grows with the bytes that arrive (the announced length allocates nothing), a frame that stalls half-way
    ends the connection, and the server caps its concurrent connections"""
    if rpc.BACKEND == "zerorpc":
        pytest.skip("real zerorpc")
    monkeypatch.setattr(rpc, "FRAME_TIMEOUT", 0.5)
    srv, url, t = serve(FakePlanner())
    port = int(url.rsplit(":", 1)[1])

    def handshaken():
        s = socket.create_connection(("127.0.0.1", port), timeout=5)
        s.sendall(RFC_GREETING + RFC_READY_DEALER)
        got = b""
        while len(got) < 64 + len(READY_ROUTER):
            got += s.recv(4096)
        return s
    # (1) a LONG frame that announces 1 TiB: the connection is closed at once
    s = handshaken()
    s.sendall(bytes([0x02]) + struct.pack("!Q", 1 << 40))
    s.settimeout(5)
    assert s.recv(16) == b""
    s.close()
    # (2) a frame that announces 200 MiB and sends 10 bytes: dropped after FRAME_TIMEOUT, and nothing of that size was allocated
    import tracemalloc
    tracemalloc.start()
    s = handshaken()
    s.sendall(bytes([0x02]) + struct.pack("!Q", 200 << 20) + b"x" * 10)
    s.settimeout(5)
    t0 = time.time()
    assert s.recv(16) == b""
    assert time.time() - t0 < 4
    peak = tracemalloc.get_traced_memory()[1]
    tracemalloc.stop()
    assert peak < (16 << 20), peak
    s.close()
    # (2b) a peer that DRIPS - a byte every 0.2 s, each well inside FRAME_TIMEOUT: the deadline is the frame's, not a recv()'s
    s = handshaken()
    s.sendall(bytes([0x02]) + struct.pack("!Q", 1 << 20))
    s.settimeout(0.05)
    t0, closed = time.time(), False
    while time.time() - t0 < 3 and not closed:
        try:
            s.sendall(b"x")
            closed = s.recv(16) == b""
        except socket.timeout:
            pass
        except ConnectionError:
            closed = True
        time.sleep(0.2)
    assert closed and time.time() - t0 < 2.5, (closed, time.time() - t0)
    s.close()
    # (3) the server still serves
    c = rpc.Client(url, timeout=5)
    assert c.add_to_env([1, 2, 3]) == 3
    c.close()
    # (4) connection cap
    monkeypatch.setattr(rpc, "MAX_CONNECTIONS", 2)
    time.sleep(0.3)
    keep = [handshaken() for _ in range(2)]
    extra = socket.create_connection(("127.0.0.1", port), timeout=5)
    extra.settimeout(3)
    try:
        assert extra.recv(16) == b""     # closed without a greeting
    except ConnectionError:
        pass
    for k in keep + [extra]:
        k.close()
    srv.stop()


def test_endpoint_validation():
    if rpc.BACKEND == "zerorpc":
        pytest.skip("real zerorpc")
    with pytest.raises(ValueError):
        rpc.Server(FakePlanner()).bind("ipc:///tmp/x")
    with pytest.raises(RuntimeError):
        rpc.Client().anything()


@pytest.mark.gpu
def test_served_hip_planner_drives_a_world_client():
    """the reference workflow end to end: MPPIisaacPlanner behind the server, a K=1 world in the client, state and
    action travelling as torch.save blobs (examples/panda/planner.py:43-48, world.py:21-48)."""
    from mppiisaac.objectives import PandaReachObjective
    from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    from mppiisaac.utils.config_store import load_config
    cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"],
                       "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14},
                      overrides={"mppi.num_samples": 512, "mppi.horizon": 12})
    planner = MPPIisaacPlanner(cfg, PandaReachObjective(cfg))
    srv, url, t = serve(planner)
    try:
        world = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1)
        world.set_actor_position_by_name([0.5, -0.4, 0.3], "goal")
        c = rpc.Client(url)
        ee = world.scene.rigid_body_index("panda", "panda_ee_tip")
        d0 = None
        for _ in range(60):
            a = bytes_to_torch(c.compute_action_tensor(torch_to_bytes(world._dof_state), torch_to_bytes(world._root_state)))
            world.apply_robot_cmd(a.to(world.device).reshape(1, -1))
            world.step()
            d = float(torch.linalg.norm(world._rigid_body_state[0, ee, 0:3] - torch.tensor([0.5, -0.4, 0.3], device=world.device)))
            d0 = d if d0 is None else d0
        assert d < 0.8 * d0                                  # the served planner steers the client's world to the goal
        assert bytes_to_torch(c.get_rollouts()).shape[1:] == (512, 3)
        c.close()
    finally:
        srv.close()
        t.join(timeout=2)
