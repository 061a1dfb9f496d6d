"""Parity of the HIP path (through the C-ABI, libmppi_hip.so) with the CPU oracle on a real MI355X.

Tolerances (fp32 device arithmetic vs the fp64 oracle), AS ASSERTED below:

* sampler, update, contact-FREE rollouts (point robot, panda and the other fixed-base arms), per sample: sampled noise 1e-6
  absolute, trajectory cost 1e-4 relative, effective perturbation 1e-6, action 1e-3 * |u_max|, joint position 1e-4 rad after a
  20-step rollout - every one of the K samples (test_full_size_properties: all 4096).
* CONTACT scenes (BASELINE configs 4 and 5) - a STATISTICAL bound plus a bound on what the controller consumes, not a per-sample
  tolerance: a rollout through contact amplifies a last-bit difference by up to 2x per substep (fp64 vs fp32 builds of the oracle
  part ways on as many samples), so over all K = 8192 samples: at the initial states >= 99.9 % within 1e-3 and max <= 1e-2; at
  the recorded closed-loop states >= 99.5 % within 1e-3 and >= 99.9 % within 1e-2; at states derived from violent rollouts
  >= 99 % / 99.9 %; EVERYWHERE the samples beyond 1e-3 carry < 1e-3 of the softmax normaliser eta (measured: 0 - they are the
  expensive, tumbling ones) and replacing the kernel's costs by the oracle's moves the nominal update by <= 1e-3 |u_max|.
  Exceptions, each stated where it is asserted: the gripper scene under the light-body law of round 6 (`held`: the weight / eta /
  update bounds and half of the samples within 1e-2; states derived from it >= 97 % / 99 %; the update bound of that scene is
  2e-2 |u_max|: test_contact_rich_states_match_oracle), and per-sample RANDOMISED actors (test_randomised_actors_per_sample: >= 98 %
  within 1e-3, at most 2 % beyond 1e-2, max < 0.1, the weight bound - pushed blocks of different sizes touch down a substep apart in
  fp32 and fp64 in a few expensive samples).
  Where along the horizon a sample leaves the oracle: tests/test_gpu_state_parity.py (per-step states of all K samples)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from mppiisaac.backend import capi
from parity_stats import agreement, fmt
from scenes import boxer_push, panda_pick, panda_reach, point_reach

pytestmark = pytest.mark.gpu


class Ctx:
    """thin test harness over the raw C-ABI (what a host binding does)."""

    def __init__(self, model, cfg, cost=None, device=0):
        self.lib = capi.load_library()
        self.model, self.cfg = model, cfg
        self.ctx = C.c_void_p()
        capi.check(self.lib, self.lib.mppi_create(C.byref(model), C.byref(cfg), device, C.byref(self.ctx)))
        if cost is not None:
            capi.check(self.lib, self.lib.mppi_set_cost(self.ctx, C.byref(cost)))
        self.K, self.H, self.nu = cfg.num_samples, cfg.horizon, cfg.nu

    def call(self, name, *args):
        capi.check(self.lib, getattr(self.lib, name)(self.ctx, *args))

    def get(self, name, shape):
        out = np.zeros(shape, np.float32)
        self.call(name, capi.fptr(out))
        return out

    def set_state(self, dof, root):
        d, r = np.ascontiguousarray(dof, np.float32), np.ascontiguousarray(root, np.float32)
        self.call("mppi_set_state", capi.fptr(d), capi.fptr(r))

    def set_U(self, U):
        u = np.ascontiguousarray(U, np.float32)
        self.call("mppi_set_nominal", capi.fptr(u))

    def close(self):
        self.lib.mppi_destroy(self.ctx)


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available()
    return capi.load_library()


@pytest.mark.parametrize("make,K,H", [(point_reach, 1024, 15), (panda_reach, 512, 20)])
def test_sample_rollout_update_match_oracle(make, K, H, lib, oracle64):
    scene, m, cfg, cost, dof, root = make(K=K, H=H)
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(0))
    eps = c.get("mppi_get_noise", (H, cfg.nu, K))
    np.testing.assert_allclose(eps, oracle64.sample(cfg), atol=1e-6)
    rng = np.random.default_rng(0)
    U0 = (0.05 * rng.normal(size=(H, cfg.nu))).astype(np.float32)
    c.set_state(dof, root)
    c.set_U(U0)
    c.call("mppi_rollout")
    S = c.get("mppi_get_costs", (K,))
    du = c.get("mppi_get_perturbations", (H, cfg.nu, K))
    So, duo, vizo = oracle64.rollout(m, cfg, cost, dof, root, U0, eps, want_viz=True)
    np.testing.assert_allclose(S, So, rtol=1e-4)
    np.testing.assert_allclose(du, duo, atol=1e-6)
    if cfg.want_rollouts:
        np.testing.assert_allclose(c.get("mppi_get_rollouts", (H, K, 3)), vizo, atol=1e-4)
    c.call("mppi_reduce", None)
    c.call("mppi_update", None, 1)
    action = c.get("mppi_get_action", (cfg.nu,))
    U1 = c.get("mppi_get_nominal", (H, cfg.nu))
    be = c.get("mppi_get_weights_stats", (2,))
    Uo, ao, beo = oracle64.update(cfg, oracle64.record(cfg, So, duo), U0)
    umax = max(abs(cfg.u_max[0]), abs(cfg.u_min[0]))
    np.testing.assert_allclose(action, ao, atol=1e-3 * umax)
    np.testing.assert_allclose(U1, Uo, atol=1e-3 * umax)
    np.testing.assert_allclose(be, beo, rtol=2e-3)
    c.close()


@pytest.mark.parametrize("mode", ["", "lane"])
def test_effort_saturated_drives_match_oracle(mode, lib, oracle64, monkeypatch):
    """The second articulated-body solve of a substep (drives held at their effort limit, quad_step's rare branch): the stock
    panda never gets there on the reach task (the implicit damping keeps the drive torque at a sixteenth of kd times the velocity
    error), so the URDF limits are cut to 4 N m / 2 N m here - then most substeps of most samples saturate one joint or several.
    Quad kernel (default) and one-lane kernel against the fp64 oracle, per sample."""
    if mode:
        monkeypatch.setenv("MPPI_ROLLOUT", mode)
    K, H = 512, 12
    scene, m, cfg, cost, dof, root = panda_reach(K=K, H=H)
    for i in range(m.n_bodies):
        m.bodies[i].effort = 4.0 if i < 4 else 2.0
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(0))
    eps = c.get("mppi_get_noise", (H, cfg.nu, K))
    rng = np.random.default_rng(1)
    U0 = (0.3 * rng.normal(size=(H, cfg.nu))).astype(np.float32)
    c.set_state(dof, root)
    c.set_U(U0)
    c.call("mppi_rollout")
    S = c.get("mppi_get_costs", (K,))
    c.close()
    So, _, _ = oracle64.rollout(m, cfg, cost, dof, root, U0, eps)
    # the limits matter: the same rollouts with the stock limits cost something else
    for i in range(m.n_bodies):
        m.bodies[i].effort = 87.0
    Sfree, _, _ = oracle64.rollout(m, cfg, cost, dof, root, U0, eps)
    assert np.mean(np.abs(Sfree - So) > 1e-3 * np.abs(So)) > 0.7
    np.testing.assert_allclose(S, So, rtol=2e-4)


def test_closed_loop_matches_oracle(lib, oracle64):
    """5 closed-loop iterations: planner (K=256) + K=1 world on the device vs the same loop on the oracle."""
    scene, m, cfg, cost, dof, root = panda_reach(K=256, H=12)
    from mppiisaac.planner.mppi import make_config
    from mppiisaac.utils.config_store import load_config
    wcfg = make_config(load_config({"defaults": [{"mppi": "panda"}]}, overrides={"mppi.num_samples": 1, "mppi.horizon": 1}).mppi)
    p, w = Ctx(m, cfg, cost), Ctx(m, wcfg)
    p.call("mppi_sample", C.c_uint32(0))
    eps = p.get("mppi_get_noise", (12, 7, 256))
    for c in (p, w):
        c.set_state(dof, root)
    w.call("mppi_sim_reset")
    q, qd, U = dof[0::2].astype(np.float64), dof[1::2].astype(np.float64), np.zeros((12, 7))
    for it in range(5):
        action = np.zeros(7, np.float32)
        p.call("mppi_command", capi.fptr(action))
        capi.check(lib, lib.mppi_world_step_from(w.ctx, p.ctx))
        capi.check(lib, lib.mppi_set_state_from_world(p.ctx, w.ctx))
        d = np.zeros(14); d[0::2], d[1::2] = q, qd
        U, ao, _ = oracle64.command(m, cfg, cost, d, root, U, eps)
        q, qd = oracle64.step(m, root, q, qd, oracle64.cmd_map(m, ao))
        np.testing.assert_allclose(action, ao, atol=2e-4)
    dev_dof = np.zeros(14, np.float32)
    p.call("mppi_get_state", capi.fptr(dev_dof), None)
    np.testing.assert_allclose(dev_dof[0::2], q, atol=1e-4)
    np.testing.assert_allclose(dev_dof[1::2], qd, atol=1e-3)
    # the fused closed-loop tail (update + world step + state feedback in one launch) gives the same loop
    p2, w2 = Ctx(m, cfg, cost), Ctx(m, wcfg)
    p2.call("mppi_sample", C.c_uint32(0))
    for c in (p2, w2):
        c.set_state(dof, root)
    w2.call("mppi_sim_reset")
    for it in range(5):
        p2.call("mppi_rollout")
        p2.call("mppi_reduce", None)
        capi.check(lib, lib.mppi_update_step_world(p2.ctx, None, 1, w2.ctx))
    dof2 = np.zeros(14, np.float32)
    p2.call("mppi_get_state", capi.fptr(dof2), None)
    np.testing.assert_allclose(dof2, dev_dof, atol=2e-5)
    np.testing.assert_allclose(p2.get("mppi_get_action", (7,)), action, atol=2e-6)
    p.close(); w.close(); p2.close(); w2.close()


def test_wait_action_sees_every_update(lib, oracle64):
    """mppi_wait_action polls the sequence number the update kernel publishes next to the action in mapped host memory:
    it must return the action of the LAST update, iteration after iteration, and agree with the stream-synchronising
    mppi_get_action."""
    scene, m, cfg, cost, dof, root = panda_reach(K=256, H=12)
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root)
    early, late = np.zeros(7, np.float32), np.zeros(7, np.float32)
    seen = []
    for _ in range(50):
        c.call("mppi_rollout"); c.call("mppi_reduce", None); c.call("mppi_update", None, 1)
        c.call("mppi_wait_action", capi.fptr(early))
        c.call("mppi_get_action", capi.fptr(late))
        np.testing.assert_array_equal(early, late)
        seen.append(early.copy())
    assert len({a.tobytes() for a in seen}) > 40          # the nominal shifts every iteration: the actions differ
    c.call("mppi_wait_action", capi.fptr(early))          # no new update: returns at once with the same action
    np.testing.assert_array_equal(early, late)
    c.close()


def test_two_shards_combine_to_single_context(lib, oracle64):
    """the N-GPU arithmetic on one GPU: two contexts own samples [0,K/2) and [K/2,K); their shard records
    combined by mppi_update must reproduce the single-context action."""
    K, H = 512, 12
    scene, m, cfg, cost, dof, root = panda_reach(K=K, H=H)
    from mppiisaac.planner.mppi import make_config
    from mppiisaac.utils.config_store import load_config
    ex = load_config({"defaults": [{"mppi": "panda"}]}, overrides={"mppi.num_samples": K, "mppi.horizon": H})
    full = Ctx(m, cfg, cost)
    full.call("mppi_sample", C.c_uint32(0)); full.set_state(dof, root)
    a_full = np.zeros(7, np.float32)
    full.call("mppi_command", capi.fptr(a_full))
    eps_full = full.get("mppi_get_noise", (H, 7, K))
    RF = lib.mppi_record_floats(full.ctx)
    records = torch.zeros((2, RF), dtype=torch.float32, device="cuda")
    shards = []
    for r in range(2):
        sc = make_config(ex.mppi, k_offset=r * K // 2, k_local=K // 2, viz_link=scene.viz_link_index())
        s = Ctx(m, sc, cost)
        s.call("mppi_sample", C.c_uint32(0)); s.set_state(dof, root)
        np.testing.assert_array_equal(s.get("mppi_get_noise", (H, 7, K // 2)), eps_full[:, :, r * K // 2:(r + 1) * K // 2])
        s.call("mppi_rollout")
        s.call("mppi_reduce", C.c_void_p(records[r].data_ptr()))
        shards.append(s)
    for s in shards:
        s.call("mppi_update", C.c_void_p(records.data_ptr()), 2)
        np.testing.assert_allclose(s.get("mppi_get_action", (7,)), a_full, atol=2e-6)
        s.close()
    full.close()


def test_full_size_properties(lib, oracle64):
    """BASELINE size K=4096, H=20 (the metric's workload): size-independent properties, and EVERY one of the 4096 samples against
    the fp64 oracle (its OpenMP loop over the samples, oracle/mppi_oracle.c orc_rollout) at the 1e-4 cost tolerance."""
    K, H = 4096, 20
    scene, m, cfg, cost, dof, root = panda_reach(K=K, H=H)
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root)
    a = np.zeros(7, np.float32)
    c.call("mppi_command", capi.fptr(a))
    S, du = c.get("mppi_get_costs", (K,)), c.get("mppi_get_perturbations", (H, 7, K))
    eps = c.get("mppi_get_noise", (H, 7, K))
    assert np.isfinite(S).all() and (S > 0).all()
    assert (np.abs(du) <= 0.2 + 1e-6).all()                       # clamp to u_min/u_max around U = 0
    np.testing.assert_array_equal(du[:, :, -1], 0.0)              # null-action sample
    # the action is the softmax-weighted mean of the effective perturbations (U0 = 0)
    w = np.exp(-(S.astype(np.float64) - S.min()) / cfg.lambda_)
    np.testing.assert_allclose(a, (du[0].astype(np.float64) * w).sum(1) / w.sum(), atol=2e-6)
    # all K samples against the oracle
    So, duo, _ = oracle64.rollout(m, cfg, cost, dof, root, np.zeros((H, 7)), eps)
    r = agreement(S, So, cfg.lambda_, du)
    print(fmt("panda_reach 4096x20", r))
    np.testing.assert_allclose(S, So, rtol=1e-4)
    np.testing.assert_allclose(du, duo, atol=1e-6)
    assert r["weight_mass_outside_1e-3"] == 0.0 and r["update_max_abs_diff"] <= 1e-5
    # determinism: same inputs -> bitwise same outputs
    c.set_U(np.zeros((H, 7)))
    c.call("mppi_rollout")
    S2 = c.get("mppi_get_costs", (K,))
    np.testing.assert_array_equal(S, S2)
    c.close()


@pytest.mark.parametrize("make,K,H,nu", [(boxer_push, 8192, 25, 2), (panda_pick, 8192, 30, 9)])
def test_full_size_properties_contact_scenes(make, K, H, nu, lib, oracle64, monkeypatch):
    """BASELINE sizes of the contact scenes (configs 4 and 5, one GPU's shard) through properties that do not need the
    oracle at full size: finite costs, clamped perturbations, the action as the weighted mean of the perturbations, bitwise
    determinism, the shared-lane (octet) kernel against the one-lane kernel on the same inputs - EVERY sample within 5 %, 99.9 %
    within 1.5 % and 99.5 % within 1e-3 - and 32 samples against the fp64 oracle (all within 1 %, nine in ten within 1e-4)."""
    scene, m, cfg, cost, dof, root = make(K=K, H=H)
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root)
    a = np.zeros(nu, np.float32)
    c.call("mppi_command", capi.fptr(a))
    S, du = c.get("mppi_get_costs", (K,)), c.get("mppi_get_perturbations", (H, nu, K))
    eps = c.get("mppi_get_noise", (H, nu, K))
    assert np.isfinite(S).all() and np.isfinite(a).all()
    umax = np.array([cfg.u_max[j] for j in range(nu)]); umin = np.array([cfg.u_min[j] for j in range(nu)])
    assert (du <= umax[None, :, None] + 1e-6).all() and (du >= umin[None, :, None] - 1e-6).all()
    np.testing.assert_array_equal(du[:, :, -1], 0.0)
    w = np.exp(-(S.astype(np.float64) - S.min()) / cfg.lambda_)
    np.testing.assert_allclose(a, (du[0].astype(np.float64) * w).sum(1) / w.sum(), atol=5e-6 * max(1.0, np.abs(umax).max()))
    c.set_U(np.zeros((H, nu)))
    c.call("mppi_rollout")
    np.testing.assert_array_equal(S, c.get("mppi_get_costs", (K,)))
    c.close()
    monkeypatch.setenv("MPPI_ROLLOUT", "lane")                      # same physics, one lane per sample
    l = Ctx(m, cfg, cost)
    l.call("mppi_sample", C.c_uint32(0)); l.set_state(dof, root); l.call("mppi_rollout")
    Sl = l.get("mppi_get_costs", (K,))
    l.close()
    # per-sample bounds (the contact force is continuous at touch-down, mppi_model_t.contact_ramp_depth; measured on MI355X,
    # tools/exp/contact_agreement.py: boxer 99.9 % within 1e-3 / max 8.5e-3, gripper scene max 1e-6)
    rel = np.abs(S - Sl) / np.abs(Sl)
    print(f"{make.__name__}: shared-lane kernel vs one-lane kernel: within 1e-3 {np.mean(rel <= 1e-3):.4f}, max {rel.max():.2e}")
    # (round 3, continuous contact law + rollout coordinates relative to the robot: every sample of the pushing scene within 1e-3
    # - worst 9.9e-4 -, gripper scene 1.1e-6; round 2 asserted 99.5 % within 1e-3 and max 5 %)
    assert (rel <= 1e-3).mean() > 0.999 and np.percentile(rel, 99.9) <= 5e-3 and rel.max() <= 2e-2
    assert np.median(S) == pytest.approx(np.median(Sl), rel=1e-4)
    if make is boxer_push:   # short tree: the default kernel has a helper wavefront per sample group; the plain octet kernel too
        info = C.create_string_buffer(256)
        monkeypatch.setenv("MPPI_ROLLOUT", "oct")
        o = Ctx(m, cfg, cost)
        o.call("mppi_kernel_info", info, 256)
        assert b"rollout=scene-oct " in info.value
        o.call("mppi_sample", C.c_uint32(0)); o.set_state(dof, root); o.call("mppi_rollout")
        rel = np.abs(o.get("mppi_get_costs", (K,)) - Sl) / np.abs(Sl)
        o.close()
        assert (rel <= 1e-3).mean() > 0.999 and np.percentile(rel, 99.9) <= 5e-3 and rel.max() <= 2e-2
    monkeypatch.delenv("MPPI_ROLLOUT")
    from mppiisaac.planner.mppi import make_config
    from mppiisaac.utils.config_store import load_config
    name = "boxer_push" if make is boxer_push else "panda_pick"
    ex = load_config({"defaults": [{"mppi": name}, {"isaacgym": "normal"}]}, overrides={"mppi.num_samples": K, "mppi.horizon": H})
    So, _, _ = oracle64.rollout(m, cfg, cost, dof, root, np.zeros((H, nu)), eps)     # ALL K samples (OpenMP over the samples)
    r = agreement(S, So, cfg.lambda_, du)
    print(fmt(f"{make.__name__} {K}x{H} initial state", r))
    # round 3 looked at 32 samples (all within 1e-4).  Over all K a few samples of the pushing scene tumble (chassis on its side: fp64
    # and fp32 part ways at 2x per substep, oracle f32 vs f64 alike - 19 of 8192 beyond 1e-3 there): asserted are 99.5 % within
    # 1e-3, 99.8 % within 1e-2, and that the samples beyond 1e-3 carry less than 1e-3 of the softmax normaliser eta and move the
    # nominal update by less than 1e-3 |u_max| (what the controller consumes)
    # measured on MI355X (profiles/r04a_gpu_tests.txt): pushing scene 99.98 % within 1e-3, max 1.4e-3 (2 samples beyond 1e-3, weight 0);
    # gripper scene every sample within 1.3e-6
    assert r["within_1e-3"] >= 0.999 and r["max"] <= 1e-2
    assert r["weight_mass_outside_1e-3"] < 1e-3 and r["update_max_abs_diff"] <= 1e-3 * np.abs(umax).max()


CLOSED_LOOP_STATES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "closed_loop_states.npz")


@pytest.mark.parametrize("make,name,K,H,nu,states", [
    (boxer_push, "boxer_push", 8192, 25, 2, ("recorded", "68_9", "13_20")),
    (panda_pick, "panda_pick", 8192, 30, 9, ("recorded", "held", "violent9", "violent20"))])
def test_contact_rich_states_match_oracle(make, name, K, H, nu, states, lib, oracle64, monkeypatch):
    """BASELINE configs 4 and 5 where the controller actually works: `recorded` = the closed-loop state after some hundred
    iterations (block against the chassis and an obstacle / gripper over the block on the table; tests/golden/
    closed_loop_states.npz, with the nominal plan U of that moment), the others = states a few steps into violent rollouts
    from there (chassis on the ground, block on the chassis, fingers in the table).  128 samples spread over the K = 8192
    against the fp64 oracle (round 3; round 4: ALL 8192 samples, with the softmax weight the disagreeing ones carry), and the
    shared-lane kernel against the one-lane kernel on all of them.
    Round 2 measured 62 % of the samples within 1e-3 at the recorded pushing state (max 25 %): the contact law was
    discontinuous (stick friction of grazing contacts, face-to-face patches beyond the explicit stability limit, joint stops)
    and the world-frame fp32 algebra lost digits two metres from the origin.  Bounds asserted here, per state:
    recorded: >= 99.5 % within 1e-3, >= 99.9 % within 1e-2; derived (violent) states: >= 99 % within 1e-3, >= 99.9 % within 1e-2;
    everywhere the samples beyond 1e-3 carry < 1e-3 of eta (measured: 0 - they are the expensive, tumbling ones)."""
    from mppiisaac.planner.mppi import make_config
    from mppiisaac.utils.config_store import load_config
    Z = np.load(CLOSED_LOOP_STATES)
    scene, m, cfg, cost, dof0, root0 = make(K=K, H=H)
    ex = load_config({"defaults": [{"mppi": name}, {"isaacgym": "normal"}]}, overrides={"mppi.num_samples": K, "mppi.horizon": H})
    problems = []   # (every state is measured and printed before anything is asserted)
    for st in states:
        dof, root = Z[f"{name}_{st}_dof"], Z[f"{name}_{st}_root"]
        U = Z[f"{name}_{st}_U"] if f"{name}_{st}_U" in Z.files else np.zeros((H, nu), np.float32)
        c = Ctx(m, cfg, cost)
        c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root); c.set_U(U); c.call("mppi_rollout")
        S, eps = c.get("mppi_get_costs", (K,)), c.get("mppi_get_noise", (H, nu, K))
        du = c.get("mppi_get_perturbations", (H, nu, K))
        c.close()
        monkeypatch.setenv("MPPI_ROLLOUT", "lane")
        l = Ctx(m, cfg, cost)
        l.call("mppi_sample", C.c_uint32(0)); l.set_state(dof, root); l.set_U(U); l.call("mppi_rollout")
        Sl = l.get("mppi_get_costs", (K,))
        l.close()
        monkeypatch.delenv("MPPI_ROLLOUT")
        assert np.isfinite(S).all() and np.isfinite(Sl).all()
        rl = np.abs(S - Sl) / np.abs(Sl)
        So, _, _ = oracle64.rollout(m, cfg, cost, dof, root, U, eps)               # ALL K samples
        r = agreement(S, So, cfg.lambda_, du)
        print(fmt(f"{name} {st}", r) + f" | shared-lane vs one-lane kernel (K samples) within 1e-3 {np.mean(rl <= 1e-3):.4f} "
              f"1e-2 {np.mean(rl <= 1e-2):.4f} max {rl.max():.1e}")
        umax = max(abs(cfg.u_max[j]) for j in range(nu))
        lanes = (float(np.mean(rl <= 1e-3)), float(np.mean(rl <= 1e-2)))
        light = name == "panda_pick"    # (round 6: the one-gram block's pairs with robot links are implicit on both bodies, robot's gains)
        if st == "held":
            # round 6, the gripper HOLDING the one-gram block 10 cm above the table, on its way (tools/record_closed_loop_states.py
            # panda_pick:40:lift).  Many rollouts from here let go of it - the sampled finger rates saturate at +-0.2 m/s - and a gram
            # that is flicked, falls and bounces parts from its fp64 twin within steps; the two fp32 kernels (shared-lane, one-lane)
            # part from each other just as often.  Asserted is what the controller consumes: the weight the disagreeing samples carry,
            # the normaliser, the nominal update - and that half of the samples still agree to 1e-2.
            # Measured (profiles/r06w_gpu_tests.txt): 89.5 % within 1e-3, 94.4 % within 1e-2; 858 samples beyond 1e-3 carrying 5e-15 of
            # eta; eta 2.8e-9; the update moves by 1.1e-9.
            want = [("weight", r["weight_mass_outside_1e-3"] < 1e-3), ("update", r["update_max_abs_diff"] <= 1e-2 * umax),
                    ("eta", r["eta_rel_err"] < 1e-2), ("half within 1e-2", r["within_1e-2"] >= 0.5)]
        elif st == "recorded":
            # where the controller works: >= 99.5 % within 1e-3, 99.9 % within 1e-2 - and the samples beyond 1e-3 carry less than
            # 1e-3 of eta; swapping the kernel's weights for the oracle's moves the nominal update by < 1e-3 |u_max|
            # measured (profiles/r04a_gpu_tests.txt, all 8192): pushing 99.87 % within 1e-3, max 4.6e-3; gripper, round 6 (`recorded` =
            # the hand closing on the block that lies on the table, 40 iterations into the task): 100 % within 1e-3 (max 1.5e-4), the
            # update moves by 6.3e-5.  The gripper scene's bound on the update is 2e-2 |u_max|: the softmax of conf/mppi/panda_pick.yaml
            # (lambda 0.05 on costs of ~240) turns a cost difference of 1e-3 ABSOLUTE - 5e-6 relative, far inside every band above -
            # into 2 % of a weight, and a state with fingers and palm ON the block (the recording of r06d) moved it by 7e-3 |u_max|.
            want = [("within", r["within_1e-3"] >= 0.995 and r["within_1e-2"] >= 0.999), ("weight", r["weight_mass_outside_1e-3"] < 1e-3),
                    ("update", r["update_max_abs_diff"] <= (2e-2 if light else 1e-3) * umax), ("lanes", lanes[0] >= 0.98 and lanes[1] >= 0.998)]
        elif light:
            # violent states DERIVED FROM `held`: the block flung out of the gripper, the arm at its joint stops (the samples with the
            # highest finite cost of that state's rollouts, 9 and 20 steps in).  Measured (r06w): 98.68 / 99.01 % within 1e-3, 99.45 /
            # 99.65 % within 1e-2, weight beyond 1e-3 <= 6e-40, update <= 9e-7; the two fp32 kernels agree on 98.8 / 98.7 %.
            want = [("within", r["within_1e-3"] >= 0.97 and r["within_1e-2"] >= 0.99), ("weight", r["weight_mass_outside_1e-3"] < 1e-3),
                    ("update", r["update_max_abs_diff"] <= 1e-2 * umax), ("lanes", lanes[0] >= 0.96 and lanes[1] >= 0.985)]
        else:
            # measured: 99.7 - 99.99 % within 1e-3, 99.96 - 100 % within 1e-2 (the tumbling samples: max 0.17 / 0.13, weight 0)
            want = [("within", r["within_1e-3"] >= 0.99 and r["within_1e-2"] >= 0.999), ("weight", r["weight_mass_outside_1e-3"] < 1e-3),
                    ("update", r["update_max_abs_diff"] <= 1e-2 * umax), ("lanes", lanes[0] >= 0.96 and lanes[1] >= 0.99)]
        problems += [f"{name} {st}: {what}" for what, ok in want if not ok]
    assert not problems, problems


def test_single_call_evaluation_is_revalidated_and_dropped_when_the_objective_drifts(lib, monkeypatch):
    """generic mode evaluates a reference-style Objective ONCE per command over the whole horizon after checking, on the first
    command, that this equals H calls on the [K]-row blocks.  An Objective whose Python-side state starts to matter LATER (here: a
    call counter that scales the cost from the 60th call on) would go stale silently: every BATCH_RECHECK-th command two row blocks
    are evaluated on their own and compared with the single call's rows (planner/mppi.py _revalidate_single) - the drift is caught,
    the full check runs, and the planner goes back to one call per horizon step with a warning."""
    import warnings
    from mppiisaac.objectives import PandaReachObjective
    from mppiisaac.planner.mppi import MPPIPlanner
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    from mppiisaac.utils.config_store import load_config

    class Drifting:
        def __init__(self):
            self.inner, self.calls = PandaReachObjective(None), 0

        def reset(self):
            pass

        def compute_cost(self, sim):
            self.calls += 1
            c = self.inner.compute_cost(sim)
            return c * (1.0 + 0.01 * self.calls) if self.calls >= 60 else c
    monkeypatch.setattr(MPPIPlanner, "BATCH_RECHECK", 4)
    H = 10
    cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"],
                       "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14},
                      overrides={"mppi.num_samples": 256, "mppi.horizon": H, "mppi.filter_u": False})
    obj = Drifting()
    planner = MPPIisaacPlanner(cfg, obj)
    planner.sim.set_actor_position_by_name([0.5, -0.4, 0.3], "goal")
    q = [0.0, -0.94, 0.0, -2.8, 0.0, 1.8675, 0.0]
    per_command = []
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        for _ in range(40):
            before = obj.calls
            planner.compute_action(q, [0.0] * 7)
            per_command.append(obj.calls - before)
    assert per_command[0] == H + 1                       # first command: the single call checked against H calls
    assert per_command[1:4] == [1, 1, 1]                 # adopted: one call per command
    assert 4 in per_command[4:12]                        # a re-validation that passes: the single call, two row blocks, the command's own call
    assert per_command[-1] == H and per_command[-5:] == [H] * 5      # after the drift: the reference's call pattern, for good
    assert any("differs from its per-step costs" in str(w.message) for w in seen)
    planner.sim.stop_sim()


def test_generic_objective_mode_equals_fused(lib):
    """Objective contract (compute_cost(sim) per horizon step, reference mppi_isaac.py:57-69) == fused kernel."""
    from mppiisaac.objectives import PandaReachObjective, PointReachObjective
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    from mppiisaac.utils.config_store import load_config
    for actors, mppi, nx, Obj, q in ((["panda_stick", "goal"], "panda", 14, PandaReachObjective, [0, -0.94, 0, -2.8, 0, 1.8675, 0]),
                                     (["point_robot", "goal"], "pointbot", 6, PointReachObjective, [0.1, 0, 0])):
        cfg = load_config({"defaults": [{"mppi": mppi}, {"isaacgym": "normal"}], "actors": actors,
                           "initial_actor_positions": [[0.0, 0.0, 0.05]], "nx": nx},
                          overrides={"mppi.num_samples": 256, "mppi.horizon": 12, "mppi.use_priors": False, "mppi.filter_u": False})

        class Generic(Obj):      # same cost, but no fused_spec -> host callback path
            fused_spec = None
        fused = MPPIisaacPlanner(cfg, Obj(cfg))
        generic = MPPIisaacPlanner(cfg, Generic(cfg))
        for pl in (fused, generic):
            pl.sim.set_actor_position_by_name([0.5, -0.4, 0.3], "goal")
        af = fused.compute_action(q, [0.0] * len(q)).numpy()
        ag = generic.compute_action(q, [0.0] * len(q)).numpy()
        Sf, Sg = fused.mppi.get_costs().numpy(), generic.mppi.get_costs().numpy()
        np.testing.assert_allclose(Sg, Sf, rtol=2e-4)
        np.testing.assert_allclose(ag, af, atol=1e-4)
        # get_rollouts: [H, K, 3] in both modes
        from mppiisaac.utils.transport import bytes_to_torch
        if fused.sim._visualize_link_present:
            rf, rg = bytes_to_torch(fused.get_rollouts()), bytes_to_torch(generic.get_rollouts()).cpu()
            assert tuple(rf.shape) == (12, 256, 3)
            np.testing.assert_allclose(rg.numpy(), rf.numpy(), atol=1e-4)


def test_generic_horizon_graph_replay_equals_eager_loop(lib, monkeypatch):
    """the generic Objective horizon is captured into a HIP graph and replayed; it must give what the reference-shaped
    eager loop gives, follow state / goal changes (they flow through sim tensors), re-capture when the weights
    change, and fall back to the eager loop for an Objective that synchronises with the host."""
    from mppiisaac.objectives import PandaReachObjective
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    from mppiisaac.utils.config_store import load_config
    from mppiisaac.utils.transport import bytes_to_torch
    cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"],
                       "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14},
                      overrides={"mppi.num_samples": 256, "mppi.horizon": 12, "mppi.use_priors": False, "mppi.filter_u": False})

    class Generic(PandaReachObjective):
        fused_spec = None
    monkeypatch.setenv("MPPI_GENERIC_BATCH", "0")          # (the whole-horizon cost call would take precedence: next test)
    graph = MPPIisaacPlanner(cfg, Generic(cfg))
    monkeypatch.setenv("MPPI_GENERIC_GRAPH", "0")
    eager = MPPIisaacPlanner(cfg, Generic(cfg))
    monkeypatch.delenv("MPPI_GENERIC_GRAPH")
    assert eager.mppi._graph_state == "off" and graph.mppi._graph_state == "auto"   # in-tree Objectives declare graph_safe
    q = np.array([0.0, -0.94, 0.0, -2.8, 0.0, 1.8675, 0.0])
    goals = [[0.5, -0.4, 0.3], [0.5, -0.4, 0.3], [0.3, 0.4, 0.5], [0.3, 0.4, 0.5]]
    for i, goal in enumerate(goals):                       # iteration 0 captures, 1.. replay; the goal moves at 2
        for pl in (graph, eager):
            pl.sim.set_actor_position_by_name(goal, "goal")
        qi = q + 0.05 * i
        ag, ae = graph.compute_action(list(qi), [0.0] * 7).numpy(), eager.compute_action(list(qi), [0.0] * 7).numpy()
        np.testing.assert_allclose(graph.mppi.get_costs().numpy(), eager.mppi.get_costs().numpy(), rtol=1e-5)
        np.testing.assert_allclose(ag, ae, atol=1e-5)
        np.testing.assert_allclose(bytes_to_torch(graph.get_rollouts()).cpu().numpy(), bytes_to_torch(eager.get_rollouts()).cpu().numpy(), atol=1e-6)
    assert graph.mppi._graph is not None and eager.mppi._graph is None
    g0 = graph.mppi._graph
    for pl in (graph, eager):                              # python-side numbers are baked in: a weight change re-captures
        pl.objective.weights["robot_ori"] = 1.5            # mutated in place, as reference examples do
    ag, ae = graph.compute_action(list(q), [0.0] * 7).numpy(), eager.compute_action(list(q), [0.0] * 7).numpy()
    assert graph.mppi._graph is not g0
    np.testing.assert_allclose(graph.mppi.get_costs().numpy(), eager.mppi.get_costs().numpy(), rtol=1e-5)

    class Syncing(PandaReachObjective):                   # .item() inside compute_cost cannot be captured
        fused_spec = None
        def compute_cost(self, sim):
            c = super().compute_cost(sim)
            return c + 0.0 * float(c[0].item())
    with pytest.warns(UserWarning, match="not graph-capturable"):
        s = MPPIisaacPlanner(cfg, Syncing(cfg))
        s.sim.set_actor_position_by_name(goals[0], "goal")
        a1 = s.compute_action(list(q), [0.0] * 7).numpy()
    assert s.mppi._graph_state == "off" and np.isfinite(a1).all()
    a2 = s.compute_action(list(q), [0.0] * 7).numpy()     # and keeps working eagerly
    assert np.isfinite(a2).all()


def test_generic_horizon_in_one_cost_call_equals_the_per_step_loop(lib, monkeypatch):
    """generic Objective mode evaluates compute_cost ONCE over an [H*K]-env view of the whole horizon after checking, on the
    first command of an Objective, that this gives the trajectory costs of the reference loop shape; it must follow state /
    goal / weight changes, keep the rollout visualisation, and refuse Objectives whose cost depends on more than the sim
    tensors (call counters) or that cannot take H*K envs."""
    from mppiisaac.objectives import PandaReachObjective, PlanarPushObjective
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    from mppiisaac.utils.config_store import load_config
    from mppiisaac.utils.transport import bytes_to_torch
    cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"],
                       "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14},
                      overrides={"mppi.num_samples": 256, "mppi.horizon": 12, "mppi.use_priors": False, "mppi.filter_u": False})

    class Generic(PandaReachObjective):
        fused_spec = None
    batched = MPPIisaacPlanner(cfg, Generic(cfg))
    monkeypatch.setenv("MPPI_GENERIC_BATCH", "0"); monkeypatch.setenv("MPPI_GENERIC_GRAPH", "0")
    eager = MPPIisaacPlanner(cfg, Generic(cfg))
    monkeypatch.delenv("MPPI_GENERIC_BATCH"); monkeypatch.delenv("MPPI_GENERIC_GRAPH")
    assert batched.mppi._batch_state == "auto" and eager.mppi._batch_state == "off"
    q = np.array([0.0, -0.94, 0.0, -2.8, 0.0, 1.8675, 0.0])
    goals = [[0.5, -0.4, 0.3], [0.5, -0.4, 0.3], [0.3, 0.4, 0.5], [0.3, 0.4, 0.5]]
    for i, goal in enumerate(goals):                       # command 0 checks (and uses the per-step loop), 1.. one call per horizon
        for pl in (batched, eager):
            pl.sim.set_actor_position_by_name(goal, "goal")
        qi = q + 0.05 * i
        ab, ae = batched.compute_action(list(qi), [0.0] * 7).numpy(), eager.compute_action(list(qi), [0.0] * 7).numpy()
        np.testing.assert_allclose(batched.mppi.get_costs().numpy(), eager.mppi.get_costs().numpy(), rtol=1e-5)
        np.testing.assert_allclose(ab, ae, atol=1e-5)
        np.testing.assert_allclose(bytes_to_torch(batched.get_rollouts()).cpu().numpy(), bytes_to_torch(eager.get_rollouts()).cpu().numpy(), atol=5e-6)
        assert batched.mppi._batch_sig[0] == "ok" and batched.mppi._graph is None
    assert batched.mppi._batch_fused is True               # the horizon came from the fused rollout kernel with the state dump
    for pl in (batched, eager):                            # a weight change is checked again
        pl.objective.weights["robot_ori"] = 1.5
    sig = batched.mppi._batch_sig
    batched.compute_action(list(q), [0.0] * 7); eager.compute_action(list(q), [0.0] * 7)
    assert batched.mppi._batch_sig != sig and batched.mppi._batch_sig[0] == "ok"
    np.testing.assert_allclose(batched.mppi.get_costs().numpy(), eager.mppi.get_costs().numpy(), rtol=1e-5)

    class Counting(PandaReachObjective):                   # the cost depends on how often it was asked: not a function of sim
        fused_spec = None
        calls = 0
        def compute_cost(self, sim):
            self.calls += 1
            return super().compute_cost(sim) * (1.0 + 0.01 * (self.calls % 7))
    with pytest.warns(UserWarning, match="differs from its per-step costs"):
        c = MPPIisaacPlanner(cfg, Counting(cfg))
        c.sim.set_actor_position_by_name(goals[0], "goal")
        a1 = c.compute_action(list(q), [0.0] * 7).numpy()
    assert c.mppi._batch_sig[0] == "no" and np.isfinite(a1).all()
    n = c.objective.calls
    c.compute_action(list(q), [0.0] * 7)                    # and stays with one call per horizon step (graph-captured or not)
    assert c.mppi._batch_sig[0] == "no"

    class FixedK(PandaReachObjective):                     # sized for K envs at construction: cannot take the H*K view
        fused_spec = None
        def compute_cost(self, sim):
            return super().compute_cost(sim) + torch.zeros(256, device=sim.device)
    with pytest.warns(UserWarning, match="cannot be evaluated over a whole horizon"):
        f = MPPIisaacPlanner(cfg, FixedK(cfg))
        f.sim.set_actor_position_by_name(goals[0], "goal")
        a1 = f.compute_action(list(q), [0.0] * 7).numpy()
    assert np.isfinite(a1).all() and np.isfinite(f.compute_action(list(q), [0.0] * 7).numpy()).all()

    # a contact scene: the pushing example with a Python Objective (floating base, per-sample actor noise, contact forces)
    pcfg = load_config({"defaults": [{"mppi": "boxer_push"}, {"isaacgym": "normal"}],
                        "actors": ["boxer", "block", "paper_obst1", "paper_obst2", "goal"], "initial_actor_positions": [[0.0, 2.5, 0.05]], "nx": 4},
                       overrides={"mppi.num_samples": 128, "mppi.horizon": 8, "mppi.use_priors": False, "mppi.filter_u": False})

    class GenericPush(PlanarPushObjective):
        fused_spec = None
    pb = MPPIisaacPlanner(pcfg, GenericPush(pcfg))
    monkeypatch.setenv("MPPI_GENERIC_BATCH", "0"); monkeypatch.setenv("MPPI_GENERIC_GRAPH", "0")
    pe = MPPIisaacPlanner(pcfg, GenericPush(pcfg))
    monkeypatch.delenv("MPPI_GENERIC_BATCH"); monkeypatch.delenv("MPPI_GENERIC_GRAPH")
    for i in range(3):
        ab = pb.compute_action([0.02 * i, 2.5, 0.0], [0.0] * 3).numpy()
        ae = pe.compute_action([0.02 * i, 2.5, 0.0], [0.0] * 3).numpy()
        # (the trajectory comes from the octet rollout kernel, the per-step loop runs the 4-lane step kernel: contact sums in
        # another order - per-sample agreement as between any two of the contact kernels - and the softmax, lambda 0.01 and
        # u_max 2, amplifies 1e-6 cost differences)
        rel = np.abs(pb.mppi.get_costs().numpy() - pe.mppi.get_costs().numpy()) / np.abs(pe.mppi.get_costs().numpy())
        assert np.median(rel) < 1e-5 and (rel <= 1e-3).mean() >= 0.98 and rel.max() <= 5e-2, (np.median(rel), rel.max())
        np.testing.assert_allclose(ab, ae, atol=2e-2)
    assert pb.mppi._batch_sig[0] == "ok" and pb.mppi._batch_fused is True
    # contexts without the dumping kernel (here: one lane per sample) simulate the horizon step by step, captured as a graph
    monkeypatch.setenv("MPPI_ROLLOUT", "lane")
    lane = MPPIisaacPlanner(cfg, Generic(cfg))
    monkeypatch.setenv("MPPI_GENERIC_BATCH", "0"); monkeypatch.setenv("MPPI_GENERIC_GRAPH", "0")
    lane_eager = MPPIisaacPlanner(cfg, Generic(cfg))
    for v in ("MPPI_ROLLOUT", "MPPI_GENERIC_BATCH", "MPPI_GENERIC_GRAPH"): monkeypatch.delenv(v)
    for i, goal in enumerate(goals[:3]):
        for pl in (lane, lane_eager):
            pl.sim.set_actor_position_by_name(goal, "goal")
        al, ae = lane.compute_action(list(q), [0.0] * 7).numpy(), lane_eager.compute_action(list(q), [0.0] * 7).numpy()
        np.testing.assert_allclose(lane.mppi.get_costs().numpy(), lane_eager.mppi.get_costs().numpy(), rtol=1e-5)
        np.testing.assert_allclose(al, ae, atol=1e-5)
    assert lane.mppi._batch_sig[0] == "ok" and lane.mppi._batch_fused is False and lane.mppi._batch_graph


def test_world_sim_matches_oracle_and_reference_layouts(lib, oracle64):
    """IsaacGymWrapper(num_envs=1): apply_robot_cmd + step and the four reference-layout tensors."""
    from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
    from mppiisaac.utils.config_store import load_config
    cfg = load_config({"defaults": [{"isaacgym": "normal"}]})
    sim = IsaacGymWrapper(cfg.isaacgym, actors=["panda_stick", "goal"], init_positions=[[0, 0, 0]], num_envs=1)
    m = sim._c_model
    dof, root = sim.scene.initial_state()
    assert sim._dof_state.shape == (1, 14) and sim._root_state.shape == (1, 2, 13)
    assert sim._rigid_body_state.shape == (1, 11, 13) and sim._net_contact_force.shape == (1, 11, 3)
    np.testing.assert_allclose(sim._dof_state[0].cpu().numpy(), dof)
    u = torch.tensor([0.1, -0.2, 0.05, 0.2, -0.1, 0.15, 0.0])
    q, qd = dof[0::2].astype(np.float64), dof[1::2].astype(np.float64)
    for _ in range(10):
        sim.apply_robot_cmd(u)
        sim.step()
        q, qd = oracle64.step(m, root, q, qd, oracle64.cmd_map(m, u.numpy()))
    got = sim.get_dof_state()[0].cpu().numpy()
    np.testing.assert_allclose(got[0::2], q, atol=1e-4)
    np.testing.assert_allclose(got[1::2], qd, atol=5e-4)
    rb, _ = oracle64.rigid_body_state(m, root, q, qd)
    np.testing.assert_allclose(sim._rigid_body_state[0].cpu().numpy(), rb, atol=1e-4)
    tip = sim.get_actor_link_by_name("panda", "panda_ee_tip")
    assert tuple(tip.shape) == (1, 13)
    np.testing.assert_allclose(sim.get_actor_position_by_name("goal")[0].cpu().numpy(), root[1, 0:3])
    assert len(sim.visualize_link_buffer) == 10 and tuple(sim.visualize_link_buffer[0].shape) == (1, 3)


def test_direct_dof_targets_equal_commands_for_unit_maps(lib):
    """set_dof_velocity_target_tensor (reference isaacgym_wrapper.py:402-403) on an arm = apply_robot_cmd; a
    differential-drive base refuses it (its command map is not the identity)"""
    from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
    from mppiisaac.utils.config_store import load_config
    cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"],
                       "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14})
    a = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1)
    b = IsaacGymWrapper(cfg.isaacgym, actors=cfg.actors, init_positions=cfg.initial_actor_positions, num_envs=1)
    u = torch.tensor([[0.1, -0.2, 0.05, 0.3, -0.1, 0.2, 0.0]], device="cuda")
    for _ in range(5):
        a.apply_robot_cmd(u); a.step()
        b.set_dof_velocity_target_tensor(u); b.step()
    np.testing.assert_array_equal(a._dof_state.cpu().numpy(), b._dof_state.cpu().numpy())
    with pytest.raises(ValueError):
        b.set_dof_actuation_force_tensor(u)                 # the arm is velocity-driven
    # by-index getters / setters of the reference (isaacgym_wrapper.py:297-397)
    g = a.scene.actor_index("goal")
    a.set_actor_position_by_actor_index([0.3, 0.2, 0.1], torch.tensor(g, device="cuda"))
    np.testing.assert_allclose(a.get_actor_position_by_name("goal")[0].cpu().numpy(), [0.3, 0.2, 0.1], atol=1e-7)
    np.testing.assert_allclose(a.get_actor_position_by_actor_index(g)[0].cpu().numpy(), [0.3, 0.2, 0.1], atol=1e-7)
    a.set_actor_velocity_by_name([0.0, 0.0, 0.5], "goal")
    assert float(a.get_actor_velocity_by_actor_index(g)[0, 2]) == pytest.approx(0.5)
    assert a.get_actor_position_by_robot_index(0).shape == (1, 3) and a.get_actor_orientation_by_robot_index(0).shape == (1, 4)
    ee = a.scene.rigid_body_index("panda", "panda_ee_tip")
    np.testing.assert_array_equal(a.get_rigid_body_by_rigid_body_index(ee).cpu().numpy(), a.get_actor_link_by_name("panda", "panda_ee_tip").cpu().numpy())
    row = torch.tensor([0.1, 0.2, 0.3, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0.0])
    a.set_root_state_tensor_by_actor_idx(row, g)
    np.testing.assert_allclose(a._root_state[0, g].cpu().numpy(), row.numpy(), atol=1e-7)
    st = a._dof_state[0].clone(); st[0] = 0.25
    a.set_actor_dof_state(st.unsqueeze(0))
    assert float(a.get_dof_state()[0, 0]) == pytest.approx(0.25)
    assert a.ostacle_velocities.shape[0] == 1 and a.draw_lines([]) is None
    cfgb = load_config({"defaults": [{"mppi": "boxer_push"}, {"isaacgym": "normal"}], "actors": ["boxer", "goal"],
                        "initial_actor_positions": [[0.0, 0.0, 0.05]], "nx": 4})
    w = IsaacGymWrapper(cfgb.isaacgym, actors=cfgb.actors, init_positions=cfgb.initial_actor_positions, num_envs=1)
    with pytest.raises(NotImplementedError):
        w.set_dof_velocity_target_tensor(torch.zeros((1, 2), device="cuda"))


def test_error_paths(lib):
    scene, m, cfg, cost, dof, root = panda_reach(K=64, H=12)
    ctx = C.c_void_p()
    bad = capi.Cost(); bad.kind = capi.COST_BOXER_PUSH
    c = Ctx(m, cfg)
    assert lib.mppi_rollout(c.ctx) == -4 and b"no fused cost" in lib.mppi_last_error()      # MPPI_ESTATE
    assert lib.mppi_set_cost(c.ctx, C.byref(bad)) == -3                                       # MPPI_EUNSUPPORTED
    c.close()
    m.drive_mode, m.drive_kp = capi.DRIVE_POSITION, -1.0                                      # a position drive needs a stiffness
    assert lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx)) == -1
    assert b"position" in lib.mppi_last_error()
    m.drive_mode, m.drive_kp = capi.DRIVE_VELOCITY, 0.0
    cfg.lambda_ = 0.0
    assert lib.mppi_create(C.byref(m), C.byref(cfg), 0, C.byref(ctx)) == -1


def test_boxer_push_rollout_matches_oracle(lib, oracle64):
    """BASELINE config 4 scene (floating diff-drive base + free block + obstacles + ground, penalty contact).
    Open-floor rollouts are smooth -> strict parity; the full-interaction case is checked through
    properties, because contact switching makes fp32/fp64 trajectories diverge chaotically."""
    K, H = 256, 12
    scene, m, cfg, cost, dof, root = boxer_push(K=K, H=H)
    root[0, 2] = 0.019
    root[scene.actor_index("block"), 0:3] = [2.5, 1.8, 0.0923]
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(0))
    eps = c.get("mppi_get_noise", (H, 2, K))
    np.testing.assert_allclose(eps, oracle64.sample(cfg), atol=2e-6)
    ext = torch.tensor(eps * 0.3, device="cuda").contiguous()       # gentle commands: stay on the open floor
    c.call("mppi_set_noise_dev", C.c_void_p(ext.data_ptr()))
    c.set_state(dof, root)
    c.call("mppi_rollout")
    S = c.get("mppi_get_costs", (K,))
    So, duo, vizo = oracle64.rollout(m, cfg, cost, dof, root, np.zeros((H, 2)), eps * 0.3, want_viz=True)
    np.testing.assert_allclose(S, So, rtol=1e-5)                                              # measured 3e-7
    np.testing.assert_allclose(c.get("mppi_get_rollouts", (H, K, 3)), vizo, atol=1e-4)        # measured 8e-6
    a = np.zeros(2, np.float32)
    c.call("mppi_reduce", None); c.call("mppi_update", None, 1)
    Uo, ao, _ = oracle64.update(cfg, oracle64.record(cfg, So, duo), np.zeros((H, 2)))
    np.testing.assert_allclose(c.get("mppi_get_action", (2,)), ao, atol=1e-5)                # measured 3e-7
    # full noise, block in front of the robot: properties
    root[scene.actor_index("block"), 0:3] = [0.0, 1.9, 0.0923]
    c.call("mppi_set_noise_dev", None)
    c.set_state(dof, root); c.set_U(np.zeros((H, 2)))
    c.call("mppi_rollout")
    S = c.get("mppi_get_costs", (K,))
    assert np.isfinite(S).all()
    So, _, _ = oracle64.rollout(m, cfg, cost, dof, root, np.zeros((H, 2)), eps)
    # same cost distribution (measured: medians 2e-7 apart, 99.2 % of the samples within 1e-3, all within 1 %; a touch-down
    # taken one substep apart in fp32 and fp64 splits the few others)
    assert np.median(S) == pytest.approx(np.median(So), rel=1e-4)
    rel = np.abs(S - So) / np.abs(So)
    print(f"pushing scene, block in front of the robot: within 1e-4 {np.mean(rel <= 1e-4):.4f} 1e-3 {np.mean(rel <= 1e-3):.4f} 1e-2 {np.mean(rel <= 1e-2):.4f} max {rel.max():.2e}")
    # round 3 (continuous contact law, rollout coordinates relative to the robot): every sample within 4.2e-4, 99.2 % within 1e-4
    # (round 2 asserted 95 % within 1e-3)
    assert (rel <= 1e-3).mean() > 0.995 and rel.max() <= 1e-2
    c.close()


def test_randomised_actors_per_sample(lib, oracle64):
    """every sample simulates its own block size / mass / friction (reference isaacgym_wrapper.py:430-475,
    seeded here). The draws are a function of the GLOBAL sample index: a shard reproduces its slice of the single-context
    costs bit for bit, and the costs follow the oracle run on the same explicitly perturbed models."""
    from mppiisaac.planner.mppi import make_config
    from mppiisaac.utils.config_store import load_config
    K, H = 256, 12
    scene, m0, cfg, cost, dof, root = boxer_push(K=K, H=H)
    root[0, 2] = 0.019
    root[scene.actor_index("block"), 0:3] = [0.0, 1.9, 0.0923]      # in front of the robot: pushed in most samples
    scene.randomize_seed = 3
    m = scene.to_c()
    eps = oracle64.sample(cfg)

    def costs(model, config):
        c = Ctx(model, config, cost)
        c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root); c.call("mppi_rollout")
        S = c.get("mppi_get_costs", (config.num_samples,))
        c.close()
        return S
    S_nom, S = costs(m0, cfg), costs(m, cfg)
    assert np.isfinite(S).all()
    assert (np.abs(S - S_nom) > 1e-3 * np.abs(S_nom)).mean() > 0.3         # the perturbed worlds differ from the nominal one
    So, _, _ = oracle64.rollout(m, cfg, cost, dof, root, np.zeros((H, 2)), eps)
    So_nom, _, _ = oracle64.rollout(m0, cfg, cost, dof, root, np.zeros((H, 2)), eps)
    # measured with the helper-wavefront kernel: 95-96 % of the samples within 1e-3 (87 % within 1e-4), medians 8e-7 apart, worst
    # sample 3.9 % - pushed blocks of different sizes touch down a substep apart in fp32 and fp64 in a few samples
    agree = (np.abs(S - So) <= 1e-3 * np.abs(So)).mean()
    print(f"randomised actors: within 1e-4 {np.mean(np.abs(S - So) <= 1e-4 * np.abs(So)):.4f} 1e-3 {agree:.4f} max {np.max(np.abs(S - So) / np.abs(So)):.2e}")
    assert np.median(S) == pytest.approx(np.median(So), rel=1e-4)
    # round 3: every sample within 3.3e-4 (round 2: 95-96 % within 1e-3, worst 3.9 %).  Round 5: the wheels and casters meet the
    # block and the obstacles too - a few samples that bump into the block or an obstacle with a wheel (cost 75 - 9000 against a
    # median of 27: softmax weight zero) part from the oracle by 1-4 %; what the controller weighs stays within 1e-3
    rel = np.abs(S - So) / np.abs(So)
    far = rel > 1e-2
    r = agreement(S, So, cfg.lambda_)
    assert agree > 0.98 and far.mean() < 0.02 and rel.max() < 0.1 and r["weight_mass_outside_1e-3"] < 1e-3, (agree, far.sum(), So[far], r)
    assert agree > (np.abs(S - So_nom) <= 1e-2 * np.abs(So_nom)).mean() + 0.1   # and it is THIS seed's worlds that it follows
    ex = load_config({"defaults": [{"mppi": "boxer_push"}, {"isaacgym": "normal"}]}, overrides={"mppi.num_samples": K, "mppi.horizon": H})
    for r in range(2):
        sc = make_config(ex.mppi, k_offset=r * K // 2, k_local=K // 2, viz_link=scene.viz_link_index())
        np.testing.assert_array_equal(costs(m, sc), S[r * K // 2:(r + 1) * K // 2])


def test_boxer_generic_mode_and_world(lib, oracle64):
    """Objective callback path and the K=1 world simulator on the contact scene: reference-layout tensors
    (root states of the moving base and block, net contact forces) against the oracle."""
    from mppiisaac.objectives import BoxerPushObjective
    from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    from mppiisaac.utils.config_store import load_config
    actors = ["boxer", "block", "paper_obst1", "paper_obst2", "goal"]
    cfg = load_config({"defaults": [{"mppi": "boxer_push"}, {"isaacgym": "normal"}], "actors": actors,
                       "initial_actor_positions": [[0.0, 2.5, 0.05]], "nx": 4},
                      overrides={"mppi.num_samples": 128, "mppi.horizon": 12, "mppi.filter_u": False})
    world = IsaacGymWrapper(cfg.isaacgym, actors=actors, init_positions=cfg.initial_actor_positions, num_envs=1)
    m = world._c_model
    dof, root = world.scene.initial_state()
    q, qd, ro = dof[0::2].astype(np.float64), dof[1::2].astype(np.float64), root.astype(np.float64)
    u = torch.tensor([0.4, 0.5])
    for _ in range(15):
        world.apply_robot_cmd(u)
        world.step()
        ro, q, qd, cfo = oracle64.scene_step(m, ro, q, qd, oracle64.cmd_map(m, u.numpy()))
    got = world._root_state[0].cpu().numpy()
    np.testing.assert_allclose(got[:, 0:7], ro[:, 0:7], atol=1e-4)      # poses after the drop + 15 steps (measured 2e-6)
    np.testing.assert_allclose(got[:, 7:13], ro[:, 7:13], atol=1e-3)    # measured 3e-5 (an approach-only damper left 1e-2 of fp32 resting jitter)
    cfg_ = world._net_contact_force[0].cpu().numpy()
    rows = [world.scene.rigid_body_index("boxer", n) for n in world.scene.link_names]
    assert cfg_[rows, 2].sum() == pytest.approx(cfo[rows, 2].sum(), rel=1e-3)
    blk = world.scene.rigid_body_index("block", "box")
    assert cfg_[blk, 2] == pytest.approx(9.8, rel=2e-3)
    rbo, _ = oracle64.rigid_body_state(m, ro, q, qd)
    np.testing.assert_allclose(world._rigid_body_state[0].cpu().numpy()[:, 0:7], rbo[:, 0:7], atol=1e-4)
    assert world.get_actor_contact_forces_by_name("paper_obst1", "box").shape == (1, 3)

    class Generic(BoxerPushObjective):
        fused_spec = None
    fused = MPPIisaacPlanner(cfg, BoxerPushObjective(cfg))
    generic = MPPIisaacPlanner(cfg, Generic(cfg))
    # the world state (settled robot) goes in through the reference's RPC entry (bytes of dof/root tensors)
    from mppiisaac.utils.transport import bytes_to_torch, torch_to_bytes
    db, rbts = torch_to_bytes(world._dof_state.cpu()), torch_to_bytes(world._root_state.cpu())
    af = bytes_to_torch(fused.compute_action_tensor(db, rbts)).numpy()
    ag = bytes_to_torch(generic.compute_action_tensor(db, rbts)).numpy()
    Sf, Sg = fused.mppi.get_costs().numpy(), generic.mppi.get_costs().numpy()
    relc = np.abs(Sf - Sg) / np.abs(Sf)
    print(f"boxer generic vs fused: costs within 1e-3 {np.mean(relc <= 1e-3):.4f}, max {relc.max():.2e}; action diff {np.abs(ag - af).max():.2e}")
    assert (relc <= 1e-3).all()            # same kernels, same arithmetic (measured: max 2.3e-4)
    np.testing.assert_allclose(ag, af, atol=1e-4)   # measured 6e-7


def test_panda_pick_rollout_matches_oracle(lib, oracle64):
    """BASELINE config 5 scene on one GPU shard (K=512 of the 8192-per-GPU workload), H=30."""
    K, H = 512, 30
    scene, m, cfg, cost, dof, root = panda_pick(K=K, H=H)
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(0))
    eps = c.get("mppi_get_noise", (H, 9, K))
    c.set_state(dof, root)
    a = np.zeros(9, np.float32)
    c.call("mppi_command", capi.fptr(a))
    S = c.get("mppi_get_costs", (K,))
    So, duo, _ = oracle64.rollout(m, cfg, cost, dof, root, np.zeros((H, 9)), eps)
    assert (np.abs(S - So) <= 1e-4 * np.abs(So)).mean() > 0.98     # measured: every sample (a touch-down can split a few)
    assert np.median(np.abs(S - So) / np.abs(So)) < 1e-5             # measured 7e-7
    Uo, ao, _ = oracle64.update(cfg, oracle64.record(cfg, So, duo), np.zeros((H, 9)))
    np.testing.assert_allclose(a, ao, atol=1e-5)                     # measured 1e-8
    c.close()


@pytest.mark.parametrize("K,H", [(1, 12), (7, 13), (100, 16), (1000, 12), (200, 15), (64, 10)])
def test_ragged_sizes(K, H, lib, oracle64):
    """sample counts that do not fill a quad-wave (16), a wavefront (64) or the XCD chunk mapping; row counts H*nu of the
    nominal that leave 0, 1, 2 or 3 rows in the last four-row group of the combine (84, 91, 112, 84, 105, 70)."""
    scene, m, cfg, cost, dof, root = panda_reach(K=K, H=H, sample_null_action=(K > 1))
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(3))
    eps = c.get("mppi_get_noise", (H, 7, K))
    np.testing.assert_allclose(eps, oracle64.sample(cfg, 3), atol=1e-6)
    c.set_state(dof, root)
    a = np.zeros(7, np.float32)
    c.call("mppi_command", capi.fptr(a))
    S = c.get("mppi_get_costs", (K,))
    So, duo, _ = oracle64.rollout(m, cfg, cost, dof, root, np.zeros((H, 7)), eps)
    np.testing.assert_allclose(S, So, rtol=1e-4)
    Uo, ao, _ = oracle64.update(cfg, oracle64.record(cfg, So, duo), np.zeros((H, 7)))
    np.testing.assert_allclose(a, ao, atol=2e-4)
    np.testing.assert_allclose(c.get("mppi_get_nominal", (H, 7)), Uo, atol=2e-4)   # every row of the shifted nominal
    c.close()


@pytest.mark.parametrize("K", [1, 7, 100, 272])
def test_ragged_sizes_contact_scene(K, lib, oracle64, monkeypatch):
    """the contact kernels that share a sample between 4 / 8 lanes (and 8 lanes + a helper wavefront) with sample counts that
    leave quads / wavefronts partly empty (and 272 = 17 wavefronts: the XCD chunk mapping falls back to the identity), and the
    one-lane kernel on the same inputs"""
    H = 10
    scene, m, cfg, cost, dof, root = boxer_push(K=K, H=H, sample_null_action=(K > 1))
    root[0, 2] = 0.019
    root[scene.actor_index("block"), 0:3] = [2.5, 1.8, 0.0923]          # open floor: smooth, strict parity
    eps = oracle64.sample(cfg) * 0.3
    So, duo, _ = oracle64.rollout(m, cfg, cost, dof, root, np.zeros((H, 2)), eps)
    _, ao, _ = oracle64.update(cfg, oracle64.record(cfg, So, duo), np.zeros((H, 2)))
    # (default for this two-wheel tree: octets with a helper wavefront that takes every other candidate pair)
    kernels = {"quad": b"rollout=scene-quad", "lane": b"rollout=scene ", "oct": b"rollout=scene-oct " if K >= 8 else b"rollout=scene-quad",
               "": b"rollout=scene-oct-pair" if K >= 8 else b"rollout=scene-quad"}
    for mode in ("quad", "lane", "oct", ""):
        if mode: monkeypatch.setenv("MPPI_ROLLOUT", mode)
        else: monkeypatch.delenv("MPPI_ROLLOUT", raising=False)
        c = Ctx(m, cfg, cost)
        ext = torch.tensor(eps, dtype=torch.float32, device="cuda").contiguous()
        c.call("mppi_set_noise_dev", C.c_void_p(ext.data_ptr()))
        c.set_state(dof, root)
        a = np.zeros(2, np.float32)
        c.call("mppi_command", capi.fptr(a))
        np.testing.assert_allclose(c.get("mppi_get_costs", (K,)), So, rtol=1e-5)
        np.testing.assert_allclose(a, ao, atol=5e-4)   # (a handful of samples: the softmax amplifies 1e-7 cost differences to 1e-4)
        info = C.create_string_buffer(256)
        c.call("mppi_kernel_info", info, 256)
        assert kernels[mode] in info.value, (mode, info.value)
        c.close()


def test_more_wave_records_than_the_combine_table_holds(lib, oracle64):
    """K = 70000 on one GPU = 4375 per-wave records, more than the 4096 rescaling factors the combine kernel caches in
    LDS: the action must still be the weighted mean over ALL samples."""
    K, H = 70000, 6
    scene, m, cfg, cost, dof, root = point_reach(K=K, H=H)
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root)
    a = np.zeros(3, np.float32)
    c.call("mppi_command", capi.fptr(a))
    S, du = c.get("mppi_get_costs", (K,)), c.get("mppi_get_perturbations", (H, 3, K))
    w = np.exp(-(S.astype(np.float64) - S.min()) / cfg.lambda_)
    np.testing.assert_allclose(a, (du[0].astype(np.float64) * w).sum(1) / w.sum(), atol=2e-6)
    assert w[65536:].sum() > 0.01 * w.sum()               # the samples beyond the table carry real weight here
    c.close()


def test_random_sampling_priors_and_param_update(lib, oracle64):
    """mppi_mode 'simple' / sampling_method 'random' (Gaussian noise drawn on the device), a prior in sample K-2
    (reference mppi_isaac.py:38-41) and update_mppi_params (:129-138) through the planner facade."""
    from mppiisaac.objectives import PandaReachObjective
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    from mppiisaac.utils.config_store import load_config
    cfg = load_config({"defaults": [{"mppi": "panda"}, {"isaacgym": "normal"}], "actors": ["panda_stick", "goal"],
                       "initial_actor_positions": [[0.0, 0.0, 0.0]], "nx": 14},
                      overrides={"mppi.num_samples": 128, "mppi.horizon": 8, "mppi.mppi_mode": "simple",
                                 "mppi.sampling_method": "random", "mppi.use_priors": True, "mppi.filter_u": False})

    class Prior:
        def compute_command(self, sim):
            return torch.full((7,), 0.05)
    pl = MPPIisaacPlanner(cfg, PandaReachObjective(cfg), prior=Prior())
    q = [0.0, -0.94, 0.0, -2.8, 0.0, 1.8675, 0.0]
    a = pl.compute_action(q, [0.0] * 7).numpy()
    sim = pl.sim
    eps = np.zeros((8, 7, 128), np.float32)
    capi.check(sim._lib, sim._lib.mppi_get_noise(sim._ctx, capi.fptr(eps)))
    assert eps.std() == pytest.approx(np.sqrt(0.1), rel=0.1)                   # N(0, noise_sigma)
    du = np.zeros((8, 7, 128), np.float32)
    capi.check(sim._lib, sim._lib.mppi_get_perturbations(sim._ctx, capi.fptr(du)))
    np.testing.assert_allclose(du[:, :, 126], 0.05, atol=1e-7)                # prior sequence in sample K-2 (U = 0)
    np.testing.assert_array_equal(du[:, :, 127], 0.0)                         # null action in sample K-1
    dof = np.zeros(14); dof[0::2] = q
    root = sim._root_state[0].cpu().numpy()
    So, duo, _ = oracle64.rollout(sim._c_model, sim._mppi_config, pl.objective.fused_spec(sim), dof, root, np.zeros((8, 7)), eps,
                                  prior=np.full((8, 7), 0.05))
    np.testing.assert_allclose(pl.mppi.get_costs().numpy(), So, rtol=1e-4)
    _, ao, _ = oracle64.update(sim._mppi_config, oracle64.record(sim._mppi_config, So, duo), np.zeros((8, 7)))
    np.testing.assert_allclose(a, ao, atol=2e-4)
    pl.update_mppi_params({"noise_sigma": (0.4 * np.eye(7)).tolist()})      # rebuilds the MPPI core with the new covariance
    pl.compute_action(q, [0.0] * 7)
    capi.check(pl.sim._lib, pl.sim._lib.mppi_get_noise(pl.sim._ctx, capi.fptr(eps)))
    assert eps.std() == pytest.approx(np.sqrt(0.4), rel=0.1)


# the shipped jackal.yaml names no wheel joints (the reference raises TypeError on it, isaacgym_wrapper.py:552-555); supply them
JACKAL_WHEELS = {"left_wheel_joints": ["front_left_wheel", "rear_left_wheel"], "right_wheel_joints": ["front_right_wheel", "rear_right_wheel"]}


@pytest.mark.parametrize("actors,init,link,nu,sigma,umax,over", [
    (["omnipanda_effort", "goal"], [[0.0, 0.0, 0.0]], "panda_hand", 12, 4.0, 10.0, None),   # effort mode, 12-DoF tree, quad kernel
    (["heijn", "goal"], [[0.0, 0.0, 0.0]], "front_link", 3, 1.0, 1.5, None),                 # holonomic base
    (["albert", "goal"], [[0.0, 0.0, 0.2]], "mmrobot_link7", 9, 0.2, 0.5, None),            # diff-drive base + arm: contact scene
    (["jackal", "goal"], [[0.0, 0.0, 0.1]], "base_link", 2, 0.5, 1.0, JACKAL_WHEELS),       # 4-wheel skid steer: contact scene
    (["panda_effort", "goal"], [[0.0, 0.0, 0.0]], "panda_link7", 7, 20.0, 40.0, None),      # effort mode, fixed base
    (["omnipanda", "goal"], [[0.0, 0.0, 0.0]], "panda_hand", 12, 0.2, 0.5, None),           # velocity mode, holonomic base + arm + gripper
    (["anymal", "goal"], [[0.0, 0.0, 0.62]], "base", 12, 1.0, 5.5, None),                     # quadruped on its feet: floating trunk, four legs
    # dof_mode "position" (reference isaacgym_wrapper.py:501-504,571-572: the command overwrites the DOF state, stiffness drive):
    (["panda_stick", "goal"], [[0.0, 0.0, 0.0]], "panda_ee_tip", 7, 0.05, 2.5, {"dof_mode": "position"}),   # quad kernel
    (["panda_gripper", "goal"], [[0.0, 0.0, 0.0]], "panda_hand", 9, 0.02, 2.5, {"dof_mode": "position"}),   # 9-body tree
    (["albert", "goal"], [[0.0, 0.0, 0.2]], "mmrobot_link7", 9, 0.05, 2.5, {"dof_mode": "position"}),       # contact scene (octet kernel)
])
def test_more_robots_rollout(actors, init, link, nu, sigma, umax, over, lib, oracle64):
    """SURVEY 8f rank 1 robots through the HIP path: reach cost on one of their links, rollouts vs the oracle."""
    from mppiisaac.planner.mppi import MPPIConfig, make_config
    from scenes import build_scene
    scene = build_scene(actors, init, robot_overrides=over)
    assert scene.nu == nu
    m = scene.to_c()
    K, H = 128, 12
    cfg = make_config(MPPIConfig(num_samples=K, horizon=H, noise_sigma=(sigma * np.eye(nu)).tolist(), lambda_=0.1, u_min=[-umax], u_max=[umax],
                                 sample_null_action=True), viz_link=scene.viz_link_index())
    cost = capi.Cost()
    cost.kind = capi.COST_PANDA_REACH
    cost.link[0] = scene.rigid_body_index(scene.robot.name, link)
    cost.actor[0] = scene.actor_index("goal")
    cost.w[0], cost.w[1] = 1.0, 0.1
    dof, root = scene.initial_state()
    root[scene.actor_index("goal"), 0:3] = [0.6, 0.3, 0.5]
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(0))
    eps = c.get("mppi_get_noise", (H, nu, K))
    c.set_state(dof, root)
    c.call("mppi_rollout")
    S = c.get("mppi_get_costs", (K,))
    So, duo, _ = oracle64.rollout(m, cfg, cost, dof, root, np.zeros((H, nu)), eps)
    assert np.isfinite(S).all()
    if oracle64.is_scene(m):   # (a touch-down taken one substep apart can split single samples: measured none, max 2.5e-7)
        assert (np.abs(S - So) <= 1e-4 * np.abs(So)).mean() > 0.97
    else:
        np.testing.assert_allclose(S, So, rtol=2e-4)
    c.close()


def test_filter_u_smooths_the_nominal(lib, oracle64):
    """filter_u (conf/mppi/panda.yaml: True): the updated nominal is U <- F (U + sum w du), then action / shift."""
    from mppiisaac.planner.mppi import savgol_matrix
    K, H = 256, 12
    scene, m, cfg, cost, dof, root = panda_reach(K=K, H=H)
    c = Ctx(m, cfg, cost)
    c.call("mppi_sample", C.c_uint32(0))
    eps = c.get("mppi_get_noise", (H, 7, K))
    F = np.ascontiguousarray(savgol_matrix(H), np.float32)
    c.call("mppi_set_filter", capi.fptr(F))
    rng = np.random.default_rng(2)
    U0 = (0.05 * rng.normal(size=(H, 7))).astype(np.float32)
    c.set_state(dof, root); c.set_U(U0)
    a = np.zeros(7, np.float32)
    c.call("mppi_command", capi.fptr(a))
    U1 = c.get("mppi_get_nominal", (H, 7))
    So, duo, _ = oracle64.rollout(m, cfg, cost, dof, root, U0, eps)
    rec = oracle64.record(cfg, So, duo)
    Unew = U0 + rec[2:].reshape(H, 7) / rec[1]
    Uf = F.astype(np.float64) @ Unew
    np.testing.assert_allclose(a, Uf[0], atol=2e-4)
    np.testing.assert_allclose(U1[:-1], Uf[1:], atol=2e-4)
    c.call("mppi_set_filter", None)
    c.set_U(U0)
    c.call("mppi_command", capi.fptr(a))
    np.testing.assert_allclose(a, Unew[0], atol=2e-4)
    c.close()


def test_dynamic_obstacles_through_compute_action(lib):
    """compute_action(q, qdot, obst=...) (reference mppi_isaac.py:71-85, isaacgym_wrapper.py:695-742): unknown
    obstacles are added as fixed spheres (simulator restart), known ones only move; an Objective that reads
    sim.obstacle_positions (as the reference's benchmark planner does) steers around them."""
    from mppiisaac.planner.mppi_isaac import MPPIisaacPlanner
    from mppiisaac.utils.config_store import load_config
    cfg = load_config({"defaults": [{"mppi": "pointbot"}, {"isaacgym": "normal"}], "actors": ["point_robot"],
                       "initial_actor_positions": [[0.0, 0.0, 0.05]], "nx": 6},
                      overrides={"mppi.num_samples": 256, "mppi.horizon": 12, "mppi.use_priors": False, "mppi.filter_u": False})

    class Objective:  # navigation + inverse-distance obstacle term (benchmarks/point_robot/.../mppi_planner_wrapper.py:17-35)
        def __init__(self):
            self.goal = torch.tensor([2.0, 0.0], device="cuda")
        def reset(self):
            pass
        def compute_cost(self, sim):
            dof = sim.get_dof_state()
            pos = torch.stack((dof[:, 0], dof[:, 2]), 1)
            nav = torch.linalg.norm(pos - self.goal, axis=1)
            obs = sim.obstacle_positions
            near = torch.sum(1 / torch.linalg.norm(obs[:, :, :2] - pos.unsqueeze(1), axis=2), axis=1) if obs.shape[1] else 0.0
            return 2.0 * nav + 1.0 * near
    pl = MPPIisaacPlanner(cfg, Objective())
    n_actors0 = len(pl.sim.env_cfg)
    obst = {"o0": {"position": [1.0, 0.05, 0.1], "velocity": [0, 0, 0], "size": [0.3]},
            "o1": {"position": [1.0, -1.5, 0.1], "velocity": [0, 0, 0], "size": [0.2]}}
    # the reference needs one call to create the actors and restart; the next call places them (:709-722)
    pl.compute_action([0.1, 0.0, 0.0], [0.0, 0.0, 0.0], obst=obst)
    assert len(pl.sim.env_cfg) == n_actors0 + 2 and pl.sim.generation == 1
    a = pl.compute_action([0.1, 0.0, 0.0], [0.0, 0.0, 0.0], obst=obst).numpy()
    assert pl.sim.generation == 1                                   # no second restart
    np.testing.assert_allclose(pl.sim.get_actor_position_by_name("sphere0")[0].cpu().numpy(), [1.0, 0.05, 0.1], atol=1e-6)
    assert pl.sim.obstacle_positions.shape == (256, 2, 3)
    assert a[0] > 0.0                                                # heads for the goal ...
    q, traj = np.array([0.1, 0.0, 0.0]), []
    qd = np.zeros(3)
    world_model = pl.sim._c_model
    from oracle.oracle import Oracle
    o = Oracle("f64")
    root = pl.sim._root_state[0].cpu().numpy().astype(np.float64)
    for _ in range(60):                                              # closed loop on the oracle's world
        a = pl.compute_action(list(q), list(qd), obst=obst).numpy()
        if o.is_scene(world_model):
            root, q, qd, _ = o.scene_step(world_model, root, q, qd, o.cmd_map(world_model, a))
        else:
            q, qd = o.step(world_model, root, q, qd, o.cmd_map(world_model, a))
        traj.append(q[:2].copy())
    traj = np.asarray(traj)
    d = np.linalg.norm(traj - np.array([1.0, 0.05]), axis=1)
    assert d.min() > 0.3 + 0.2 - 0.05                                # ... around the obstacle (radius 0.3, robot radius 0.2)
    assert np.linalg.norm(traj[-1] - np.array([2.0, 0.0])) < 0.6


def test_body_force_reference_test_shape(lib, oracle64):
    """The reference's only test (mppiisaac/planner/tests/test_isaacgym_wrapper.py:11-35, `test_body_force`): boxer + wall in
    600 envs, the same (0.2, 0) command everywhere, 200 steps; it asserts the DOF tensor shape and that identical envs
    report identical contact forces.  (Its `sim.net_cf` / `sim.dof_state` attributes are `_net_contact_force` /
    `_dof_state` in the wrapper of the same commit.)"""
    from mppiisaac.planner.isaacgym_wrapper import IsaacGymWrapper
    from mppiisaac.utils.config_store import load_config
    cfg = load_config({"defaults": [{"isaacgym": "normal"}]})
    num_envs = 600
    sim = IsaacGymWrapper(cfg.isaacgym, actors=["boxer", "wall"], init_positions=[[0.0, 0.0, 0.5]], num_envs=num_envs)
    assert sim._dof_state.size() == torch.Size([num_envs, 4])
    cmd = torch.Tensor([0.2, 0.0]).repeat(num_envs, 1)
    assert cmd.size() == torch.Size([num_envs, 2])
    sim.apply_robot_cmd(cmd)
    for _ in range(200):
        sim.step()
    cf = sim._net_contact_force.cpu()
    assert torch.isfinite(cf).all()
    assert (cf == cf[0:1]).all()                                   # every env identical, bit for bit
    assert (sim._root_state.cpu() == sim._root_state[0:1].cpu()).all()
    base = sim._root_state[0, 0].cpu()
    travelled = float(torch.linalg.norm(base[0:2]))
    assert base[2] < 0.05 and 1.8 < travelled < 2.05               # dropped onto its wheels, then 0.2 m/s for ~10 s
    assert cf[0].abs().sum() > 1.0                                 # standing on the ground: wheel / caster forces
    # the same 200 steps through the oracle
    m = sim._c_model
    dof, root = sim.scene.initial_state()
    q, qd, root = dof[0::2].astype(np.float64), dof[1::2].astype(np.float64), root.astype(np.float64)
    target = oracle64.cmd_map(m, np.array([0.2, 0.0]))
    for _ in range(200):
        root, q, qd, _ = oracle64.scene_step(m, root, q, qd, target)
    np.testing.assert_allclose(sim._root_state[0, 0, 0:3].cpu().numpy(), root[0, 0:3], atol=2e-2)


@pytest.mark.gpu
def test_rollout_trajectory_states_match_oracle_stepping(lib, oracle64):
    """mppi_rollout_trajectory + mppi_materialise_trajectory through the raw C-ABI: row t*K + k of the reference-layout tensors
    is env k after horizon step t of the SAME rollout the fused kernel runs - checked against the oracle stepping each sample
    with its clamped controls (contact-free arm: 1e-5; pushing scene with floating base, free block and contact forces)."""
    # contact-free: panda reach
    K, H = 64, 6
    scene, m, cfg, cost, dof, root = panda_reach(K=K, H=H)
    c = Ctx(m, cfg)                                              # (no fused cost: the generic-mode context)
    c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root)
    A, B, n = len(scene.env_cfg), scene.n_rb, scene.n_dof
    f32 = dict(dtype=torch.float32, device="cuda")
    T = {"dof": torch.zeros((H * K, 2 * n), **f32), "root": torch.zeros((H * K, A, 13), **f32),
         "rb": torch.zeros((H * K, B, 13), **f32), "cf": torch.zeros((H * K, B, 3), **f32)}
    ptr = lambda t: C.c_void_p(t.data_ptr())
    c.call("mppi_rollout_trajectory")
    c.call("mppi_materialise_trajectory", ptr(T["dof"]), ptr(T["root"]), ptr(T["rb"]), ptr(T["cf"]))
    du, eps = c.get("mppi_get_perturbations", (H, 7, K)), c.get("mppi_get_noise", (H, 7, K))
    S = c.get("mppi_get_costs", (K,))
    dofs, rbs = T["dof"].cpu().numpy().reshape(H, K, 2 * n), T["rb"].cpu().numpy().reshape(H, K, B, 13)
    assert not T["cf"].any()
    for k in (0, 17, K - 2, K - 1):
        q, qd = dof[0::2].astype(np.float64), dof[1::2].astype(np.float64)
        for t in range(H):
            u = du[t, :, k].astype(np.float64)                  # U = 0: the applied control is the effective perturbation
            q, qd = oracle64.step(m, root, q, qd, oracle64.cmd_map(m, u))
            np.testing.assert_allclose(dofs[t, k, 0::2], q, atol=1e-5)
            np.testing.assert_allclose(dofs[t, k, 1::2], qd, atol=2e-4)
            rbo, _ = oracle64.rigid_body_state(m, root, q, qd)
            np.testing.assert_allclose(rbs[t, k, :, 0:3], rbo[:, 0:3], atol=2e-5)
    assert np.abs(du[:, :, K - 1]).max() == 0.0                  # the null-action sample
    # S holds the control cost only (no fused cost): lambda * sum_t U^T Sigma^-1 du = 0 for U = 0
    np.testing.assert_allclose(S, 0.0, atol=1e-7)
    info = C.create_string_buffer(256)
    c.call("mppi_kernel_info", info, 256)
    c.close()
    # contact scene: boxer push (helper-wavefront kernel with the dump), per-sample actor noise off
    K, H = 64, 5
    scene, m, cfg, cost, dof, root = boxer_push(K=K, H=H)
    c = Ctx(m, cfg)
    c.call("mppi_sample", C.c_uint32(0)); c.set_state(dof, root)
    A, B, n = len(scene.env_cfg), scene.n_rb, scene.n_dof
    T = {"dof": torch.zeros((H * K, 2 * n), **f32), "root": torch.zeros((H * K, A, 13), **f32),
         "rb": torch.zeros((H * K, B, 13), **f32), "cf": torch.zeros((H * K, B, 3), **f32)}
    c.call("mppi_rollout_trajectory")
    c.call("mppi_materialise_trajectory", ptr(T["dof"]), ptr(T["root"]), ptr(T["rb"]), ptr(T["cf"]))
    du = c.get("mppi_get_perturbations", (H, 2, K))
    dofs, roots = T["dof"].cpu().numpy().reshape(H, K, 2 * n), T["root"].cpu().numpy().reshape(H, K, A, 13)
    cfs = T["cf"].cpu().numpy().reshape(H, K, B, 3)
    for k in (0, 9, K - 2):
        r, q, qd = root.astype(np.float64), dof[0::2].astype(np.float64), dof[1::2].astype(np.float64)
        for t in range(H):
            r, q, qd, cfo = oracle64.scene_step(m, r, q, qd, oracle64.cmd_map(m, du[t, :, k].astype(np.float64)))
            np.testing.assert_allclose(roots[t, k, :, 0:7], r[:, 0:7], atol=1e-4)
            np.testing.assert_allclose(dofs[t, k, 0::2], q, atol=1e-3)
            # (N; touch-down loads of 5-17 kN here; a contact that starts a rounding error earlier or later shows up as ~1 N,
            # the law being continuous at zero depth)
            np.testing.assert_allclose(cfs[t, k], cfo, rtol=5e-3, atol=5.0)
    c.close()
    # contexts on the one-lane kernels have no dumping kernel: refused with a reason, the mppi_sim_* steps remain
    os.environ["MPPI_ROLLOUT"] = "lane"
    try:
        scene, m, cfg, cost, dof, root = panda_reach(K=64, H=6)
        c = Ctx(m, cfg)
        assert lib.mppi_rollout_trajectory(c.ctx) == -3 and b"trajectory" in lib.mppi_last_error()   # MPPI_EUNSUPPORTED
        c.close()
    finally:
        del os.environ["MPPI_ROLLOUT"]


def _golden_cases():
    from test_golden_boundary import EXAMPLE_SCENES
    return sorted(EXAMPLE_SCENES)


@pytest.mark.parametrize("case", _golden_cases())
def test_golden_objective_inputs_through_the_hip_cost_program(case, lib):
    """The reference's Objectives straight against the HIP cost path: tests/golden/objective_costs.json holds what the gym getters
    returned (seeded random link rows, actor rows, contact forces, DOF states) and what each example planner's compute_cost made
    of them (tools/make_golden.py imports the reference's examples/<x>/planner.py).  Those inputs go through mppi_eval_cost -
    the cost-program interpreter the rollout kernels run (program_cost_with), on the device, fed with the given rows instead of
    the kernel's kinematics - and must give the reference's costs.  (Until round 4 the chain was golden -> oracle on the CPU,
    oracle -> HIP on the GPU.)"""
    import mppiisaac.objectives as objectives
    from mppiisaac.planner.mppi import MPPIConfig, make_config
    from scenes import build_scene
    from test_golden_boundary import EXAMPLE_SCENES, OBJECTIVES, gold
    g = gold("objective_costs.json")[case]
    scene = build_scene(EXAMPLE_SCENES[case], [[0.0, 0.0, 0.05]])
    m = scene.to_c()
    obj = getattr(objectives, OBJECTIVES[case])(None)

    class Sim:
        pass
    sim = Sim()
    sim.scene = scene
    spec = obj.program_spec(sim)
    cfg = make_config(MPPIConfig(num_samples=64, horizon=4, noise_sigma=np.eye(m.nu).tolist()))
    c = Ctx(m, cfg, spec)
    n, nd = len(g["cost"]), scene.n_dof
    dof, root = np.zeros((n, 2 * nd), np.float32), np.zeros((n, m.n_actors, 13), np.float32)
    rb, cf = np.zeros((n, m.n_rb, 13), np.float32), np.zeros((n, m.n_rb, 3), np.float32)
    rb[:, :, 6] = 1.0   # (rows the objective never asks for: identity quaternions)
    root[:, :, 6] = 1.0
    for key, val in g["inputs"].items():
        val = np.asarray(val, np.float32)
        kind, *names = key.split(":")
        if kind == "link":
            rb[:, scene.rigid_body_index(*names)] = val
        elif kind == "contact":
            cf[:, scene.rigid_body_index(*names)] = val
        elif kind == "dof_state":
            dof[:, :] = val[:, :2 * nd]
        else:
            col = {"position": slice(0, 3), "orientation": slice(3, 7), "velocity": slice(7, 10)}[kind]
            root[:, scene.actor_index(names[0]), col] = val
    # box / sphere actors are rigid bodies too: a program that names their body reads the actor's root row
    out = np.zeros(n, np.float32)
    c.call("mppi_eval_cost", n, capi.fptr(dof), capi.fptr(root), capi.fptr(rb), capi.fptr(cf), capi.fptr(out))
    c.close()
    want = np.asarray(g["cost"])
    print(f"{case}: HIP cost program vs reference Objective on {n} golden envs: max rel {np.max(np.abs(out - want) / np.maximum(np.abs(want), 1e-6)):.1e}")
    np.testing.assert_allclose(out, want, rtol=2e-6, atol=2e-6)   # fp32 interpreter, hardware sqrt / rcp (1 ulp) vs the reference's torch fp32: measured 1.6e-7
