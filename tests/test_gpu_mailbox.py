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
            np.testing.assert_allclose(s.get("mppi_get_action", (nu,)), a_full, atol=3e-6)
            np.testing.assert_allclose(s.get("mppi_get_nominal", (H, nu)), U_full, atol=3e-6)
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
            np.testing.assert_allclose(a, a_full, atol=3e-6)
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
