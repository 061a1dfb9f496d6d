"""The library-owned exchange of the shard records (mppi_mailbox_* / mppi_exchange, include/mppi_hip.h; SURVEY.md 8e: the
1-hop mailbox all-gather over xGMI) on ONE device: G shard contexts on G HIP streams stand in for the G GPUs of a node -
device pointers for peer pointers, everything else (publish into every inbox, release, flags, bounded poll, acquire, gather)
is what runs across GPUs.  Compared with the single-context result and with the RCCL-style path (records gathered by the
caller, mppi_update on all of them)."""
import ctypes as C

import numpy as np
import pytest
import torch

from mppiisaac.backend import capi
from mppiisaac.planner.mppi import make_config
from mppiisaac.utils.config_store import load_config
from scenes import boxer_push, panda_reach
from test_gpu_parity import Ctx

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available()
    return capi.load_library()


def dev_to_host(ptr, n_floats):
    out = np.zeros(n_floats, np.float32)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), ptr, n_floats * 4, 2) == 0   # hipMemcpyDeviceToHost
    return out


def shard_contexts(make, name, K, H, G, lib):
    scene, m, cfg, cost, dof, root = make(K=K, H=H)
    ex = load_config({"defaults": [{"mppi": name}, {"isaacgym": "normal"}]}, overrides={"mppi.num_samples": K, "mppi.horizon": H})
    streams = [torch.cuda.Stream() for _ in range(G)]
    shards = []
    for r in range(G):
        sc = make_config(ex.mppi, k_offset=r * K // G, k_local=K // G, viz_link=scene.viz_link_index())
        s = Ctx(m, sc, cost)
        s.call("mppi_set_stream", C.c_void_p(streams[r].cuda_stream))
        s.call("mppi_sample", C.c_uint32(0)); s.set_state(dof, root)
        s.call("mppi_mailbox_create", r, G)
        shards.append(s)
    ptrs = []
    for s in shards:
        p, n = C.c_void_p(), C.c_size_t()
        s.call("mppi_mailbox_ptr", C.byref(p), C.byref(n))
        assert n.value > 0
        ptrs.append(p)
    for s in shards:
        for r in range(G):
            s.call("mppi_mailbox_set_peer", r, ptrs[r])
    return scene, m, cfg, cost, dof, root, shards, streams


@pytest.mark.parametrize("make,name,K,H,G", [(panda_reach, "panda", 4096, 20, 2), (panda_reach, "panda", 8192, 20, 8), (boxer_push, "boxer_push", 4096, 12, 4)])
def test_mailbox_exchange_equals_single_context(make, name, K, H, G, lib):
    scene, m, cfg, cost, dof, root, shards, streams = shard_contexts(make, name, K, H, G, lib)
    nu = cfg.nu
    full = Ctx(m, cfg, cost)
    full.call("mppi_sample", C.c_uint32(0)); full.set_state(dof, root)
    gathered = []
    for s in shards:
        p, n = C.c_void_p(), C.c_int()
        s.call("mppi_mailbox_gathered", C.byref(p), C.byref(n))
        assert n.value == G * max(1, lib.mppi_shard_record_count(s.ctx))
        gathered.append((p, n.value))
    RF = lib.mppi_record_floats(full.ctx)
    for it in range(3):                                  # several iterations: sequence numbers, both slot parities, shifted nominal
        a_full = np.zeros(nu, np.float32)
        full.call("mppi_command", capi.fptr(a_full))
        U_full = full.get("mppi_get_nominal", (H, nu))
        # every shard: rollout -> exchange -> update, each on its own stream, enqueued back to back (no host synchronisation
        # between the ranks: the polls wait on the device for the peers' kernels on the other streams)
        order = list(reversed(shards)) if it == 1 else shards   # (launch order must not matter)
        if G == 2:
            for s in order:
                s.call("mppi_rollout")
                s.call("mppi_exchange")                          # publish + wait back to back: the peer's stream has its own queue
        else:
            # one thread drives all ranks: every publish is enqueued before any wait (streams may share hardware queues, and
            # a waiting kernel would hold up a peer's publish queued behind it - across processes / GPUs there is no such order)
            for s in order:
                s.call("mppi_rollout")
                s.call("mppi_exchange_publish")
            for s in order:
                s.call("mppi_exchange_wait")
        for s, (p, n) in zip(shards, gathered):
            s.call("mppi_update", p, n)
        for s in shards:
            late = C.c_int(-1)
            s.call("mppi_exchange_status", C.byref(late))
            assert late.value == 0
            np.testing.assert_allclose(s.get("mppi_get_action", (nu,)), a_full, rtol=3e-6, atol=3e-6)   # (fp32 sums of the shard records in another order: measured 1.4e-6 relative on an action of 3.03)
            np.testing.assert_allclose(s.get("mppi_get_nominal", (H, nu)), U_full, rtol=3e-6, atol=3e-6)
        # all ranks hold the same gathered records, bit for bit
        torch.cuda.synchronize()
        ref = dev_to_host(gathered[0][0], gathered[0][1] * RF)
        assert np.isfinite(ref).all() and ref.reshape(-1, RF)[:, 1].sum() > 0
        for p, n in gathered[1:]:
            np.testing.assert_array_equal(dev_to_host(p, n * RF), ref)
    for s in shards:
        s.close()
    full.close()


def test_exchange_then_tail_on_a_contact_scene_equals_single_context(lib):
    """the same call on a contact scene (8 folded records per shard, K=1 world stepped by its own kernel): the library takes
    mppi_exchange and mppi_update_step_world in turn - two shard contexts with a world each against one context, closed loop"""
    K, H, G = 2048, 8, 2
    scene, m, cfg, cost, dof, root, shards, streams = shard_contexts(boxer_push, "boxer_push", K, H, G, lib)
    wcfg = make_config(load_config({"defaults": [{"mppi": "boxer_push"}]}, overrides={"mppi.num_samples": 1, "mppi.horizon": 1}).mppi)
    nu = cfg.nu
    full, wfull = Ctx(m, cfg, cost), Ctx(m, wcfg)
    full.call("mppi_sample", C.c_uint32(0))
    worlds = []
    for i in range(G):
        w = Ctx(m, wcfg)
        w.call("mppi_set_stream", C.c_void_p(streams[i].cuda_stream))
        worlds.append(w)
    for c in [full, wfull] + worlds:
        c.set_state(dof, root)
    for w in [wfull] + worlds:
        w.call("mppi_sim_reset")
    torch.cuda.synchronize()
    a_full, a = np.zeros(nu, np.float32), np.zeros(nu, np.float32)
    for it in range(4):
        full.call("mppi_rollout")
        capi.check(lib, lib.mppi_update_step_world(full.ctx, None, 1, wfull.ctx))
        full.call("mppi_get_action", capi.fptr(a_full))
        for s in shards:
            s.call("mppi_rollout")
        for s, w in zip(shards, worlds):
            capi.check(lib, lib.mppi_exchange_update_step_world(s.ctx, w.ctx))
        for s in shards:
            s.call("mppi_get_action", capi.fptr(a))
            # (the summation order of the records differs between one context and two shards: 1e-7 on the first action - and the
            # closed loop through a contact scene amplifies that from iteration to iteration)
            np.testing.assert_allclose(a, a_full, atol=3e-6 if it == 0 else 2e-3)
    assert np.abs(a_full).max() > 0
    for c in shards + worlds + [full, wfull]:
        c.close()


@pytest.mark.parametrize("G", [2, 4])
def test_exchange_fused_into_the_closed_loop_tail_equals_single_context(G, lib):
    """mppi_exchange_update_step_world: for the contact-free scenes the exchange is the head of the combine + world kernel (reduce
    the wavefront records to the shard record, publish, bounded wait, combine the ranks' records, step the K=1 world, feed its
    state back) - ONE launch per rank and iteration.  G shard contexts with a world each against one context with its world,
    closed loop: the actions of later iterations depend on the fed-back state, so equal actions mean the whole tail worked."""
    K, H = 2048, 12
    scene, m, cfg, cost, dof, root, shards, streams = shard_contexts(panda_reach, "panda", K, H, G, lib)
    wcfg = make_config(load_config({"defaults": [{"mppi": "panda"}]}, overrides={"mppi.num_samples": 1, "mppi.horizon": 1}).mppi)
    nu = cfg.nu
    full, wfull = Ctx(m, cfg, cost), Ctx(m, wcfg)
    full.call("mppi_sample", C.c_uint32(0))
    worlds = []
    for i, s in enumerate(shards):
        w = Ctx(m, wcfg)
        w.call("mppi_set_stream", C.c_void_p(streams[i].cuda_stream))
        worlds.append(w)
    for c in [full, wfull] + worlds:
        c.set_state(dof, root)
    for w in [wfull] + worlds:
        w.call("mppi_sim_reset")
    torch.cuda.synchronize()
    a_full, a = np.zeros(nu, np.float32), np.zeros(nu, np.float32)
    for it in range(5):
        full.call("mppi_rollout")
        capi.check(lib, lib.mppi_update_step_world(full.ctx, None, 1, wfull.ctx))
        full.call("mppi_wait_action", capi.fptr(a_full))
        for s in shards:                                  # (rollouts first: every rank's tail kernel waits for all publishes)
            s.call("mppi_rollout")
        for s, w in zip(shards, worlds):
            capi.check(lib, lib.mppi_exchange_update_step_world(s.ctx, w.ctx))
        for s in shards:
            s.call("mppi_wait_action", capi.fptr(a))
            late = C.c_int(-1)
            s.call("mppi_exchange_status", C.byref(late))
            assert late.value == 0
            np.testing.assert_allclose(a, a_full, rtol=3e-6, atol=3e-6)
        assert np.abs(a_full).max() > 0
    info = C.create_string_buffer(256)
    shards[0].call("mppi_kernel_info", info, 256)
    for c in shards + worlds + [full, wfull]:
        c.close()


def test_mailbox_wait_is_bounded(lib):
    """a rank whose peer never publishes does not hang the device: the poll gives up after ~2 s and says so"""
    scene, m, cfg, cost, dof, root, shards, streams = shard_contexts(panda_reach, "panda", 512, 8, 2, lib)
    s = shards[0]
    s.call("mppi_rollout")
    s.call("mppi_exchange")          # rank 1 never runs
    late = C.c_int(0)
    s.call("mppi_exchange_status", C.byref(late))
    assert late.value == 1
    # read-and-clear (ABI 8): the late wait is reported once - a transient stall does not mark every later iteration late
    s.call("mppi_exchange_status", C.byref(late))
    assert late.value == 0
    for c in shards:
        c.close()


def test_mailbox_ipc_handle_roundtrip(lib):
    """the handle another process would open (hipIpcGetMemHandle of the inbox) can be produced; opening it is a cross-process
    operation and is exercised by the two-process test below"""
    scene, m, cfg, cost, dof, root, shards, streams = shard_contexts(panda_reach, "panda", 512, 8, 2, lib)
    h = (C.c_ubyte * 64)()
    shards[0].call("mppi_mailbox_ipc_handle", h)
    assert any(bytes(h))
    for c in shards:
        c.close()


def test_late_rank_is_neutral_in_the_update(lib):
    """a peer that never publishes: the wait gives up (bounded), the status word says so, and the late rank's slots reach the
    update as NEUTRAL records (eta = 0) - the action is the one this rank's own records give, not one mixed with whatever an
    old or half-written slot held (RCCL would hang or raise here)."""
    K, H, G = 1024, 8, 2
    scene, m, cfg, cost, dof, root, shards, streams = shard_contexts(panda_reach, "panda", K, H, G, lib)
    nu, s = cfg.nu, shards[0]
    RF = lib.mppi_record_floats(s.ctx)
    p, n = C.c_void_p(), C.c_int()
    s.call("mppi_mailbox_gathered", C.byref(p), C.byref(n))
    # poison rank 1's slots first, as a stale iteration would leave them: publish from rank 1 ONCE, with absurd records
    shards[1].call("mppi_rollout"); shards[0].call("mppi_rollout")
    shards[1].call("mppi_exchange_publish"); shards[0].call("mppi_exchange")     # iteration 1: both publish, rank 0 gathers
    s.call("mppi_update", p, n.value)
    # iteration 2 and 3: rank 1 is silent (slot parity 0 was never written by rank 1; parity 1 holds its iteration-1 records)
    for _ in range(2):
        s.call("mppi_rollout")
        s.call("mppi_exchange")
        late = C.c_int(0)
        s.call("mppi_exchange_status", C.byref(late))
        assert late.value == 1
        torch.cuda.synchronize()
        rec = dev_to_host(p, n.value * RF).reshape(n.value, RF)
        assert rec[0, 1] > 0 and rec[1, 1] == 0.0                     # own record live, the late rank's neutral
        U = s.get("mppi_get_nominal", (H, nu))
        own, _ = C.c_void_p(), None
        capi.check(lib, lib.mppi_record_dev(s.ctx, C.byref(own)))     # this shard's own record
        torch.cuda.synchronize()
        r0 = dev_to_host(own, RF)
        np.testing.assert_allclose(rec[0], r0, rtol=1e-6)
        s.call("mppi_update", p, n.value)
        a = s.get("mppi_get_action", (nu,))
        np.testing.assert_allclose(a, U[0] + r0[2:2 + nu] / r0[1], atol=3e-6)   # the update of the ranks that did publish
    for c in shards:
        c.close()


def test_mailbox_create_and_open_refusals_leave_a_usable_context(lib):
    """negative paths of the set-up: bad arguments, a second create, a stale / foreign / garbage IPC handle - each is an error code
    with a message, nothing half-built stays behind, and the context keeps working through the plain path afterwards"""
    K, H = 512, 8
    scene, m, cfg, cost, dof, root = panda_reach(K=K, H=H)
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root)
    fine, nrec, nr = C.c_int(), C.c_int(), C.c_int()
    assert lib.mppi_mailbox_info(c.ctx, C.byref(fine), C.byref(nrec), C.byref(nr)) == capi.MPPI_ESTATE     # no mailbox yet
    for rank, n in ((2, 2), (-1, 2), (0, 0), (0, 65)):
        assert lib.mppi_mailbox_create(c.ctx, rank, n) == capi.MPPI_EINVAL
        assert lib.mppi_mailbox_info(c.ctx, C.byref(fine), C.byref(nrec), C.byref(nr)) == capi.MPPI_ESTATE  # nothing left behind
    assert lib.mppi_exchange(c.ctx) == capi.MPPI_ESTATE and lib.mppi_exchange_wait(c.ctx) == capi.MPPI_ESTATE
    c.call("mppi_mailbox_create", 0, 2)
    assert lib.mppi_mailbox_create(c.ctx, 0, 2) == capi.MPPI_ESTATE                                      # a second create is refused ...
    c.call("mppi_mailbox_info", C.byref(fine), C.byref(nrec), C.byref(nr))                                # ... and the first one is intact
    assert (nrec.value, nr.value) == (1, 2)
    assert lib.mppi_exchange(c.ctx) == capi.MPPI_ESTATE and b"not connected" in lib.mppi_last_error()    # peer 1 was never connected
    for blob in (bytes(64), bytes(range(64)), b"\xff" * 64):                                            # garbage handles: an error, not a crash
        assert lib.mppi_mailbox_open(c.ctx, 1, (C.c_ubyte * 64).from_buffer_copy(blob)) == capi.MPPI_EHIP
    assert lib.mppi_mailbox_open(c.ctx, 5, (C.c_ubyte * 64)()) == capi.MPPI_EINVAL                        # rank out of range
    assert lib.mppi_mailbox_set_peer(c.ctx, 1, None) == capi.MPPI_EINVAL
    a = np.zeros(cfg.nu, np.float32)
    c.call("mppi_command", capi.fptr(a))                                                                  # the context still plans
    assert np.isfinite(a).all() and np.abs(a).max() > 0
    c.close()


def test_mailbox_sized_for_folded_records_serves_generic_mode(lib):
    """a contact scene whose rollout folds 8 records per shard sizes its mailbox for 8 - and a generic Objective
    (host-side costs: nothing is folded) then has ONE reduced record to publish.  It goes out with neutral padding; two shards
    through the mailbox give the single context's update."""
    K, H, G = 4096, 6, 2
    scene, m, cfg, cost, dof, root, shards, streams = shard_contexts(boxer_push, "boxer_push", K, H, G, lib)
    nu = cfg.nu
    assert lib.mppi_shard_record_count(shards[0].ctx) == 8
    full = Ctx(m, cfg, cost)
    full.call("mppi_sample", C.c_uint32(0)); full.set_state(dof, root)
    host_cost = (1.0 + torch.sin(0.37 * torch.arange(K, dtype=torch.float32, device="cuda"))).contiguous()   # stands in for compute_cost(sim)

    def generic_horizon(c, costs):     # the reference's loop shape (mppi_isaac.py:57-69) with a precomputed cost tensor
        c.call("mppi_sim_reset")
        for t in range(H):
            c.call("mppi_sim_step_horizon", t)
            c.call("mppi_sim_accumulate_cost", t, C.c_void_p(costs.data_ptr()))
        c.call("mppi_sim_finish")
    torch.cuda.synchronize()
    generic_horizon(full, host_cost)
    full.call("mppi_reduce", None); full.call("mppi_update", None, 1)
    a_full, U_full = full.get("mppi_get_action", (nu,)), full.get("mppi_get_nominal", (H, nu))
    torch.cuda.synchronize()
    parts = [host_cost[r * K // G:(r + 1) * K // G].contiguous() for r in range(G)]
    for s_, part in zip(shards, parts):
        generic_horizon(s_, part)
    torch.cuda.synchronize()
    for s_ in shards:
        s_.call("mppi_exchange_publish")
    for s_ in shards:
        s_.call("mppi_exchange_wait")
    for s_ in shards:
        p, n = C.c_void_p(), C.c_int()
        s_.call("mppi_mailbox_gathered", C.byref(p), C.byref(n))
        assert n.value == 8 * G
        s_.call("mppi_update", p, n.value)
        late = C.c_int(-1)
        s_.call("mppi_exchange_status", C.byref(late))
        assert late.value == 0
        np.testing.assert_allclose(s_.get("mppi_get_action", (nu,)), a_full, rtol=3e-6, atol=3e-6)
        np.testing.assert_allclose(s_.get("mppi_get_nominal", (H, nu)), U_full, rtol=3e-6, atol=3e-6)
    for c in shards + [full]:
        c.close()
