/*
 * mppi_oracle.c - CPU restatement of the MPPI rollout hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / timed CPU baseline - never from the product path.
 *
 * What it restates (paths relative to the reference tree tud-airlab/mppi-isaac):
 *   - the control-iteration loop shape: sample -> for t<H: apply_robot_cmd -> step -> cost,
 *     then exp-weights and nominal update     mppiisaac/planner/mppi_isaac.py:57-69,107-113
 *   - command scatter + diff-drive IK          mppiisaac/planner/isaacgym_wrapper.py:510-572
 *   - drive gains, gravity, dt/substeps        isaacgym_wrapper.py:21-39,491-507
 *   - state layouts (dof interleaved, root/rigid-body 13-vectors, quat xyzw)  :186-199
 *   - example stage costs                      examples/panda/planner.py:22-40,
 *                                              benchmarks/point_robot/mppi_planner/mppi_planner_wrapper.py:17-35
 *
 * PARITY STATUS.  The two engines the reference delegates to are absent from its tree and
 * cannot be built or imported here:
 *   - Isaac Gym 1.0rc4 / PhysX (closed binary; pyproject.toml:16)    -> dynamics: PARITY UNPINNED
 *   - mppi_torch @75e17e87 (un-vendored git dep; poetry.lock:1272-1293) -> MPPI arithmetic: PARITY UNPINNED
 *   - pytorch3d 0.3.0 rotation conversions (poetry.lock:2027-2029): restated from the published
 *     algorithm, known-answer tested against scipy.spatial.transform (tests/test_oracle_kat.py).
 * The dynamics follow SURVEY.md section B (Featherstone articulated-body algorithm, implicit
 * velocity-level joint drive, semi-implicit Euler), the MPPI arithmetic SURVEY.md section A.
 * The boundary logic that IS importable from the reference (command scatter, diff-drive IK,
 * state packing, quaternion_to_yaw) is pinned by tests/golden/ (tools/make_golden.py), and so are
 * the stage costs: orc_cost reproduces Objective.compute_cost of examples/{panda,boxer_push,
 * panda_pick}/planner.py on recorded simulator answers (tests/golden/objective_costs.json).
 *
 * Formulation: textbook body-coordinate spatial algebra with explicit 6x6 matrices
 * (Featherstone, Rigid Body Dynamics Algorithms, 2008, Table 7.1) - deliberately different
 * from the HIP kernel's world-frame structured form, so the two check each other.
 *
 * Build: make -C oracle   (gcc, -DREAL=double -> liboracle_f64.so, -DREAL=float -> liboracle_f32.so)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mppi_hip.h"

#ifndef REAL
#define REAL double
#endif
typedef REAL real;

#define NBMAX MPPI_MAX_BODIES
#define NBASEMAX (1 + MPPI_MAX_EXTRA_BASES)

/* ------------------------------------------------------------------ small linear algebra */
static void m3_mul(const real *A, const real *B, real *C) { /* C = A B */
    real T[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, T, sizeof T);
}
static void m3_vec(const real *A, const real *x, real *y) {
    real t[3];
    for (int i = 0; i < 3; i++) t[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
    y[0] = t[0]; y[1] = t[1]; y[2] = t[2];
}
static void m3_tvec(const real *A, const real *x, real *y) { /* y = A^T x */
    real t[3];
    for (int i = 0; i < 3; i++) t[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
    y[0] = t[0]; y[1] = t[1]; y[2] = t[2];
}
static void cross3(const real *a, const real *b, real *c) {
    real t0 = a[1] * b[2] - a[2] * b[1], t1 = a[2] * b[0] - a[0] * b[2], t2 = a[0] * b[1] - a[1] * b[0];
    c[0] = t0; c[1] = t1; c[2] = t2;
}
static void skew3(const real *v, real *S) {
    S[0] = 0; S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2]; S[4] = 0; S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
/* Rodrigues rotation about unit axis a by angle q */
static void rot_axis(const real *a, real q, real *R) {
    real c = (real)cos((double)q), s = (real)sin((double)q), v = 1 - c;
    R[0] = c + a[0] * a[0] * v;        R[1] = a[0] * a[1] * v - a[2] * s; R[2] = a[0] * a[2] * v + a[1] * s;
    R[3] = a[1] * a[0] * v + a[2] * s; R[4] = c + a[1] * a[1] * v;        R[5] = a[1] * a[2] * v - a[0] * s;
    R[6] = a[2] * a[0] * v - a[1] * s; R[7] = a[2] * a[1] * v + a[0] * s; R[8] = c + a[2] * a[2] * v;
}
/* quaternion xyzw (reference root/rigid-body layout, isaacgym_wrapper.py:186-195) -> R */
static void quat_to_R(const real *q, real *R) {
    real x = q[0], y = q[1], z = q[2], w = q[3];
    real n = x * x + y * y + z * z + w * w;
    real s = n > 0 ? 2 / n : 0;
    R[0] = 1 - s * (y * y + z * z); R[1] = s * (x * y - z * w);     R[2] = s * (x * z + y * w);
    R[3] = s * (x * y + z * w);     R[4] = 1 - s * (x * x + z * z); R[5] = s * (y * z - x * w);
    R[6] = s * (x * z - y * w);     R[7] = s * (y * z + x * w);     R[8] = 1 - s * (x * x + y * y);
}
/* R -> quaternion xyzw, canonical sign w >= 0 (Shepperd) */
static void R_to_quat(const real *R, real *q) {
    real tr = R[0] + R[4] + R[8];
    real x, y, z, w;
    if (tr > 0) {
        real s = (real)sqrt((double)(tr + 1)) * 2;
        w = s / 4; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        real s = (real)sqrt((double)(1 + R[0] - R[4] - R[8])) * 2;
        w = (R[7] - R[5]) / s; x = s / 4; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        real s = (real)sqrt((double)(1 + R[4] - R[0] - R[8])) * 2;
        w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = s / 4; z = (R[5] + R[7]) / s;
    } else {
        real s = (real)sqrt((double)(1 + R[8] - R[0] - R[4])) * 2;
        w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = s / 4;
    }
    if (w < 0) { x = -x; y = -y; z = -z; w = -w; }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

/* 6x6, row-major; spatial vectors are [angular(3); linear(3)] */
static void m6_vec(const real *A, const real *x, real *y) {
    real t[6];
    for (int i = 0; i < 6; i++) { t[i] = 0; for (int j = 0; j < 6; j++) t[i] += A[6 * i + j] * x[j]; }
    memcpy(y, t, sizeof t);
}
static void m6_tvec(const real *A, const real *x, real *y) {
    real t[6];
    for (int i = 0; i < 6; i++) { t[i] = 0; for (int j = 0; j < 6; j++) t[i] += A[6 * j + i] * x[j]; }
    memcpy(y, t, sizeof t);
}
/* C = X^T A X */
static void m6_congruence(const real *X, const real *A, real *C) {
    real T[36];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { real s = 0; for (int k = 0; k < 6; k++) s += A[6 * i + k] * X[6 * k + j]; T[6 * i + j] = s; }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { real s = 0; for (int k = 0; k < 6; k++) s += X[6 * k + i] * T[6 * k + j]; C[6 * i + j] = s; }
}
/* Plucker motion transform parent->child for x_parent = R x_child + p:
 *   X = [ E 0 ; -E p^x  E ],  E = R^T   (RBDA eq. 2.24 with r = p) */
static void plucker(const real *R, const real *p, real *X) {
    real E[9], px[9], Epx[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) E[3 * i + j] = R[3 * j + i];
    skew3(p, px);
    m3_mul(E, px, Epx);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            X[6 * i + j] = E[3 * i + j];         X[6 * i + 3 + j] = 0;
            X[6 * (i + 3) + j] = -Epx[3 * i + j]; X[6 * (i + 3) + 3 + j] = E[3 * i + j];
        }
}
/* rigid-body spatial inertia about the frame origin (RBDA eq. 2.63): [Io  h^x ; -h^x  m 1] */
static void rigid_inertia(real m, const real *h, const real *Io6, real *I) {
    real hx[9];
    skew3(h, hx);
    real Io[9] = {Io6[0], Io6[1], Io6[2], Io6[1], Io6[3], Io6[4], Io6[2], Io6[4], Io6[5]};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            I[6 * i + j] = Io[3 * i + j];          I[6 * i + 3 + j] = hx[3 * i + j];
            I[6 * (i + 3) + j] = -hx[3 * i + j];   I[6 * (i + 3) + 3 + j] = (i == j) ? m : 0;
        }
}
static void crm(const real *v, const real *m, real *out) { /* v x m (motion) */
    real a[3], b[3], c[3];
    cross3(v, m, a); cross3(v, m + 3, b); cross3(v + 3, m, c);
    out[0] = a[0]; out[1] = a[1]; out[2] = a[2];
    out[3] = b[0] + c[0]; out[4] = b[1] + c[1]; out[5] = b[2] + c[2];
}
static void crf(const real *v, const real *f, real *out) { /* v x* f (force) */
    real a[3], b[3], c[3];
    cross3(v, f, a); cross3(v + 3, f + 3, b); cross3(v, f + 3, c);
    out[0] = a[0] + b[0]; out[1] = a[1] + b[1]; out[2] = a[2] + b[2];
    out[3] = c[0]; out[4] = c[1]; out[5] = c[2];
}

/* ------------------------------------------------------------------ kinematics of one state */
typedef struct {
    real Rj[NBMAX][9], pj[NBMAX][3]; /* child->parent transform of each body            */
    real Rw[NBMAX][9], pw[NBMAX][3]; /* body->world                                      */
    real Rb[NBASEMAX][9], pb[NBASEMAX][3]; /* base r -> world (r > 0: the further moving-base robots of an env, ABI 7) */
    real vb[NBASEMAX][6];            /* base spatial velocity, base coordinates          */
    real X[NBMAX][36];               /* Plucker parent->child                            */
    real S[NBMAX][6];
    real v[NBMAX][6];                /* spatial velocity, body coordinates               */
    real c[NBMAX][6];
} kin_t;

/* the moving bases of an env (mppi_hip.h, ABI 7): base 0 = robot_actor, base r > 0 = extra_base_*[r - 1]; a body's parent / a
 * link's or shape's body index -1 - r names base r */
static int n_bases(const mppi_model_t *m) { return 1 + m->n_extra_bases; }
static int base_actor(const mppi_model_t *m, int r) { return r == 0 ? m->robot_actor : m->extra_base_actor[r - 1]; }
static real base_mass_of(const mppi_model_t *m, int r) { return (real)(r == 0 ? m->base_mass : m->extra_base_mass[r - 1]); }
static const double *base_h_of(const mppi_model_t *m, int r) { return r == 0 ? m->base_h : m->extra_base_h[r - 1]; }
static const double *base_Io_of(const mppi_model_t *m, int r) { return r == 0 ? m->base_Io : m->extra_base_Io[r - 1]; }
static int base_of_body(const mppi_model_t *m, int i) {
    while (m->bodies[i].parent >= 0) i = m->bodies[i].parent;
    return -1 - m->bodies[i].parent;
}

static void kinematics(const mppi_model_t *m, const real *root, const real *q, const real *qd, kin_t *k) {
    for (int r = 0; r < n_bases(m); r++) {
        const real *rs = root + 13 * base_actor(m, r);
        k->pb[r][0] = rs[0]; k->pb[r][1] = rs[1]; k->pb[r][2] = rs[2];
        quat_to_R(rs + 3, k->Rb[r]);
        /* base spatial velocity in base coordinates (zero for a fixed base) */
        for (int j = 0; j < 6; j++) k->vb[r][j] = 0;
        if (!m->actors[base_actor(m, r)].fixed) { m3_tvec(k->Rb[r], rs + 10, k->vb[r]); m3_tvec(k->Rb[r], rs + 7, k->vb[r] + 3); }
    }
    for (int i = 0; i < m->n_bodies; i++) {
        const mppi_body_t *b = &m->bodies[i];
        real Rt[9], pt[3], ax[3];
        for (int j = 0; j < 9; j++) Rt[j] = (real)b->R_tree[j];
        for (int j = 0; j < 3; j++) { pt[j] = (real)b->p_tree[j]; ax[j] = (real)b->axis[j]; }
        if (b->jtype == MPPI_JOINT_REVOLUTE) {
            real Rq[9];
            rot_axis(ax, q[i], Rq);
            m3_mul(Rt, Rq, k->Rj[i]);
            for (int j = 0; j < 3; j++) k->pj[i][j] = pt[j];
            for (int j = 0; j < 3; j++) { k->S[i][j] = ax[j]; k->S[i][3 + j] = 0; }
        } else {
            real d[3];
            memcpy(k->Rj[i], Rt, sizeof Rt);
            m3_vec(Rt, ax, d);
            for (int j = 0; j < 3; j++) k->pj[i][j] = pt[j] + d[j] * q[i];
            for (int j = 0; j < 3; j++) { k->S[i][j] = 0; k->S[i][3 + j] = ax[j]; }
        }
        plucker(k->Rj[i], k->pj[i], k->X[i]);
        const real *Rp = b->parent < 0 ? k->Rb[-1 - b->parent] : k->Rw[b->parent];
        const real *pp = b->parent < 0 ? k->pb[-1 - b->parent] : k->pw[b->parent];
        real t[3];
        m3_mul(Rp, k->Rj[i], k->Rw[i]);
        m3_vec(Rp, k->pj[i], t);
        for (int j = 0; j < 3; j++) k->pw[i][j] = pp[j] + t[j];
        /* v_i = X v_parent + S qd ; c_i = v_i x (S qd)    (base is fixed: v_base = 0) */
        real vj[6];
        for (int j = 0; j < 6; j++) vj[j] = k->S[i][j] * qd[i];
        m6_vec(k->X[i], b->parent < 0 ? k->vb[-1 - b->parent] : k->v[b->parent], k->v[i]);
        for (int j = 0; j < 6; j++) k->v[i][j] += vj[j];
        crm(k->v[i], vj, k->c[i]);
    }
}

/* Articulated-body algorithm (RBDA Table 7.1) with a per-joint implicit velocity-level drive:
 *   tau_i = tau_exp[i] - kdh[i]*qdd_i   <=>   d_i += kdh[i]      
 * Fixed base.  Gravity enters as the fictitious base acceleration a_0 = -g. */
static void aba_solve(const mppi_model_t *m, const kin_t *k, const real *tau_exp, const real *kdh, real *qdd) {
    int n = m->n_bodies;
    real IA[NBMAX][36], pA[NBMAX][6], U[NBMAX][6], d[NBMAX], u[NBMAX], a[NBMAX][6];
    for (int i = 0; i < n; i++) {
        const mppi_body_t *b = &m->bodies[i];
        real h[3] = {(real)b->h[0], (real)b->h[1], (real)b->h[2]};
        real Io[6];
        for (int j = 0; j < 6; j++) Io[j] = (real)b->Io[j];
        rigid_inertia((real)b->mass, h, Io, IA[i]);
        real Iv[6];
        m6_vec(IA[i], k->v[i], Iv);
        crf(k->v[i], Iv, pA[i]);
    }
    for (int i = n - 1; i >= 0; i--) {
        m6_vec(IA[i], k->S[i], U[i]);
        real sd = 0, sp = 0;
        for (int j = 0; j < 6; j++) { sd += k->S[i][j] * U[i][j]; sp += k->S[i][j] * pA[i][j]; }
        d[i] = sd + kdh[i];
        u[i] = tau_exp[i] - sp;
        int par = m->bodies[i].parent;
        if (par >= 0) {
            real Ia[36], pa[6], t6[6], Xt[36];
            for (int r = 0; r < 6; r++) for (int cc = 0; cc < 6; cc++) Ia[6 * r + cc] = IA[i][6 * r + cc] - U[i][r] * U[i][cc] / d[i];
            m6_vec(Ia, k->c[i], t6);
            for (int j = 0; j < 6; j++) pa[j] = pA[i][j] + t6[j] + U[i][j] * (u[i] / d[i]);
            m6_congruence(k->X[i], Ia, Xt);
            for (int j = 0; j < 36; j++) IA[par][j] += Xt[j];
            m6_tvec(k->X[i], pa, t6);
            for (int j = 0; j < 6; j++) pA[par][j] += t6[j];
        }
    }
    real a0[6] = {0, 0, 0, 0, 0, 0};
    if (m->actors[m->robot_actor].gravity) {
        real g[3] = {(real)m->gravity[0], (real)m->gravity[1], (real)m->gravity[2]}, gb[3];
        m3_tvec(k->Rb[0], g, gb);   /* (fixed-base trees and forests: ONE base frame, mppi_hip.h; several bases move and are scenes) */
        a0[3] = -gb[0]; a0[4] = -gb[1]; a0[5] = -gb[2];
    }
    for (int i = 0; i < n; i++) {
        int par = m->bodies[i].parent;
        real ap[6];
        m6_vec(k->X[i], par < 0 ? a0 : a[par], ap);
        real ua = 0;
        for (int j = 0; j < 6; j++) { ap[j] += k->c[i][j]; }
        for (int j = 0; j < 6; j++) ua += U[i][j] * ap[j];
        qdd[i] = (u[i] - ua) / d[i];
        for (int j = 0; j < 6; j++) a[i][j] = ap[j] + k->S[i][j] * qdd[i];
    }
}

/* plain forward dynamics qdd = ABA(q, qd, tau) - exported for the known-answer tests */
void orc_forward_dynamics(const mppi_model_t *m, const real *root, const real *q, const real *qd, const real *tau, real *qdd) {
    kin_t k;
    real zero[NBMAX] = {0};
    kinematics(m, root, q, qd, &k);
    aba_solve(m, &k, tau, zero, qdd);
}

/* apply_robot_cmd: control vector u [nu] -> per-DOF drive target (isaacgym_wrapper.py:524-572) */
void orc_cmd_map(const mppi_model_t *m, const real *u, real *target) {
    for (int i = 0; i < m->n_bodies; i++)
        target[i] = (real)m->cmd_coef[i][0] * u[m->cmd_col[i][0]] + (real)m->cmd_coef[i][1] * u[m->cmd_col[i][1]];
}

/* diagnostics (tools/exp/saturation_stats.py): per substep, the set of joints whose drive saturated (bit i) and the sign
 * of the clamped force (bit 16 + i); NULL = off */
static uint32_t *g_sat_log = NULL;
static int g_sat_n = 0;
#pragma omp threadprivate(g_sat_log, g_sat_n)
/* One simulator step of dt = substeps * h (IsaacGymWrapper.step, isaacgym_wrapper.py:639-645). */
/* Inelastic joint limit: the position is clamped and the velocity becomes the displacement that actually happened over the
 * step, (q_new - q_old) / h - never pointing back out of the range.  (Setting the velocity to zero at the stop is a jump: a
 * joint that reaches its limit one substep earlier or later - 1e-9 rad decide - differs by its full speed for that substep.) */
static void joint_limit(real qold, real *q, real *qd, real lower, real upper, real h) {
    if (*q < lower) { *q = lower; real ve = (lower - qold) / h; *qd = ve < 0 ? ve : 0; }
    if (*q > upper) { *q = upper; real ve = (upper - qold) / h; *qd = ve > 0 ? ve : 0; }
}

/* Drives (reference isaacgym_wrapper.py:491-507): velocity tau = kd (target - qd), effort tau = target - kd qd, position
 * tau = kp (target - q) - kd qd - all implicit in the velocity: with qd+ = qd + h qdd and q+ = q + h qd+ the position drive is
 * kp (target - q) - (kd + h kp) qd+ , i.e. the same form with the damping kde = kd + h kp.  Position mode also TELEPORTS:
 * apply_robot_cmd overwrites the DOF state with the command (:571-572) before the step. */
static void drive_teleport(const mppi_model_t *m, real *q, real *qd, const real *target) {
    if (m->drive_mode != MPPI_DRIVE_POSITION) return;
    for (int i = 0; i < m->n_bodies; i++) { q[i] = target[i]; qd[i] = 0; }
}
static real drive_damping(const mppi_model_t *m, real h) {
    return (real)m->drive_kd + (m->drive_mode == MPPI_DRIVE_POSITION ? h * (real)m->drive_kp : 0);
}
void orc_step(const mppi_model_t *m, const real *root, real *q, real *qd, const real *target) {
    int n = m->n_bodies;
    real h = (real)(m->dt / m->substeps), kd = drive_damping(m, h), kp = m->drive_mode == MPPI_DRIVE_POSITION ? (real)m->drive_kp : 0;
    drive_teleport(m, q, qd, target);
    for (int s = 0; s < m->substeps; s++) {
        kin_t k;
        kinematics(m, root, q, qd, &k);
        real ff[NBMAX], vs[NBMAX], tau[NBMAX], kdh[NBMAX], qdd[NBMAX];
        for (int i = 0; i < n; i++) {
            ff[i] = m->drive_mode == MPPI_DRIVE_EFFORT ? target[i] : (m->drive_mode == MPPI_DRIVE_POSITION ? kp * (target[i] - q[i]) : 0);
            vs[i] = m->drive_mode == MPPI_DRIVE_VELOCITY ? target[i] : 0;

            tau[i] = ff[i] + kd * (vs[i] - qd[i]);
            kdh[i] = kd * h;
        }
        aba_solve(m, &k, tau, kdh, qdd);
        /* drive-force clamp (URDF <limit effort>): joints whose implicit drive force exceeds the
         * limit are re-solved with the constant saturated force (one re-solve) */
        int any = 0;
        uint32_t satmask = 0;
        for (int i = 0; i < n; i++) {
            real lim = (real)m->bodies[i].effort;
            real tt = ff[i] + kd * (vs[i] - qd[i] - h * qdd[i]);
            if (lim > 0 && (real)fabs((double)tt) > lim) { any = 1; tau[i] = tt > 0 ? lim : -lim; kdh[i] = 0; satmask |= (1u << i) | (tt > 0 ? (1u << (16 + i)) : 0u); }
        }
        if (g_sat_log) g_sat_log[g_sat_n++] = satmask;
        if (any) aba_solve(m, &k, tau, kdh, qdd);
        for (int i = 0; i < n; i++) {
            const mppi_body_t *b = &m->bodies[i];
            real vmax = (real)b->velocity;
            qd[i] += h * qdd[i];
            if (vmax > 0) { if (qd[i] > vmax) qd[i] = vmax; if (qd[i] < -vmax) qd[i] = -vmax; }
            const real qold = q[i];
            q[i] += h * qd[i];
            if (b->limited) joint_limit(qold, &q[i], &qd[i], (real)b->lower, (real)b->upper, h);
        }
    }
}

/* rigid_body_state rows [n_rb][13] (pos, quat xyzw, linvel, angvel; world frame) and
 * net_contact_force rows [n_rb][3] for ONE env (isaacgym_wrapper.py:193-199). */
void orc_rigid_body_state(const mppi_model_t *m, const real *root, const real *q, const real *qd, real *rb, real *cf) {
    kin_t k;
    kinematics(m, root, q, qd, &k);
    for (int a = 0; a < m->n_actors; a++) {
        const mppi_actor_t *A = &m->actors[a];
        if (a != m->robot_actor) { /* box / sphere: its single body is the root body */
            /* (the further robots of a forest: their links' rows are written with the first robot's, the rows follow one another) */
            if (A->type != MPPI_ACTOR_ROBOT) memcpy(rb + 13 * A->first_rb, root + 13 * a, 13 * sizeof(real));
            continue;
        }
        for (int l = 0; l < m->n_links; l++) {
            const mppi_link_t *L = &m->links[l];
            real Rl[9], pl[3], Rw[9], pw[3], t[3], quat[4], wv[3] = {0, 0, 0}, lv[3] = {0, 0, 0};
            for (int j = 0; j < 9; j++) Rl[j] = (real)L->R[j];
            for (int j = 0; j < 3; j++) pl[j] = (real)L->p[j];
            const real *Rbw = L->body < 0 ? k.Rb[-1 - L->body] : k.Rw[L->body];
            const real *pbw = L->body < 0 ? k.pb[-1 - L->body] : k.pw[L->body];
            m3_mul(Rbw, Rl, Rw);
            m3_vec(Rbw, pl, t);
            for (int j = 0; j < 3; j++) pw[j] = pbw[j] + t[j];
            { /* velocity of the link origin: R_w (v + w x p_l) */
                const real *v = L->body >= 0 ? k.v[L->body] : k.vb[-1 - L->body];
                real wxp[3], vl[3];
                cross3(v, pl, wxp);
                for (int j = 0; j < 3; j++) vl[j] = v[3 + j] + wxp[j];
                m3_vec(Rbw, vl, lv);
                m3_vec(Rbw, v, wv);
            }
            R_to_quat(Rw, quat);
            real *o = rb + 13 * (A->first_rb + l);
            o[0] = pw[0]; o[1] = pw[1]; o[2] = pw[2];
            o[3] = quat[0]; o[4] = quat[1]; o[5] = quat[2]; o[6] = quat[3];
            o[7] = lv[0]; o[8] = lv[1]; o[9] = lv[2];
            o[10] = wv[0]; o[11] = wv[1]; o[12] = wv[2];
        }
    }
    if (cf) for (int j = 0; j < 3 * m->n_rb; j++) cf[j] = 0; /* no contact model in this scope row */
}

/* ------------------------------------------------------------------ contact scenes
 * Floating-base robots, free rigid bodies and penalty contact.
 * The reference delegates all of this to PhysX (isaacgym_wrapper.py:21-39 sim params, :429-508 actor and
 * shape properties, isaacgym_utils.py:61-68 ground plane); PARITY UNPINNED - this is the build-normative
 * model, checked by physics known-answer tests (tests/test_scene_kat.py).
 * Formulation: dense 6x6 spatial algebra in WORLD coordinates about the world origin, double precision. */
#define NFMAX (NBMAX + NBASEMAX + MPPI_MAX_FREE)

typedef struct {
    real R[9], p[3];   /* pose */
    real v[6];         /* spatial velocity (w, vO) */
    real f[6];         /* external wrench accumulator */
    real C[36];        /* implicit damping accumulator */
    real mass;         /* mass of the owning actor (contact scaling) */
} frame_t;

typedef struct {
    int is_scene, floating, n_free, free_actor[MPPI_MAX_FREE];
    int light[MPPI_MAX_ACTORS];  /* the actor is a LIGHT free body (see scene_contacts) */
    int nb;                      /* bases of the forest (1 + n_extra_bases): dynamic frames n_bodies .. n_bodies + nb - 1 */
    real robot_mass[NBASEMAX];   /* mass of the tree that hangs off base r (contact gains scale with the reacting robot's mass) */
} scene_info_t;

static void scene_info(const mppi_model_t *m, scene_info_t *si) {
    si->floating = !m->actors[m->robot_actor].fixed;
    si->n_free = 0;
    for (int a = 0; a < m->n_actors; a++)
        if (m->actors[a].type != MPPI_ACTOR_ROBOT && !m->actors[a].fixed && si->n_free < MPPI_MAX_FREE) si->free_actor[si->n_free++] = a;
    si->nb = n_bases(m);
    for (int r = 0; r < si->nb; r++) si->robot_mass[r] = base_mass_of(m, r);
    for (int i = 0; i < m->n_bodies; i++) si->robot_mass[base_of_body(m, i)] += (real)m->bodies[i].mass;
    si->is_scene = si->floating || si->n_free > 0 || m->n_pairs > 0;
    /* light bodies (see scene_contacts): a free actor of at most MPPI_LIGHT_BODY_MASS that the robot outweighs MPPI_LIGHT_BODY_RATIO
     * times and that has no candidate pair with another free actor */
    for (int a = 0; a < MPPI_MAX_ACTORS; a++) si->light[a] = 0;
    if (!(m->contact_flags & MPPI_CONTACT_EXPLICIT_LIGHT))
        for (int f = 0; f < si->n_free; f++) {
            const int a = si->free_actor[f];
            int ok = m->actors[a].mass <= MPPI_LIGHT_BODY_MASS && (double)si->robot_mass[0] >= MPPI_LIGHT_BODY_RATIO * m->actors[a].mass;
            for (int ip = 0; ip < m->n_pairs && ok; ip++) {
                if (m->pairs[ip].b < 0) continue;
                const int aa = m->shapes[m->pairs[ip].a].actor, ab = m->shapes[m->pairs[ip].b].actor;
                if (aa != a && ab != a) continue;
                const int other = aa == a ? ab : aa;
                for (int g = 0; g < si->n_free; g++) if (si->free_actor[g] == other) ok = 0;
            }
            si->light[a] = ok;
            if (ok) break;   /* (one light body per scene: what the kernels carry) */
        }
}
int orc_is_scene(const mppi_model_t *m) { scene_info_t si; scene_info(m, &si); return si.is_scene; }

/* C6 += J^T (b 1 + (a-b) n n^T) J with J = [-[p]x 1]: viscous element acting at point p */
static void damping_add(real *C, const real *p, const real *n, real a, real b) {
    real J[18]; /* 3x6 */
    real px[9];
    skew3(p, px);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { J[6 * i + j] = -px[3 * i + j]; J[6 * i + 3 + j] = (i == j) ? 1 : 0; }
    real Cl[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Cl[3 * i + j] = (i == j ? b : 0) + (a - b) * n[i] * n[j];
    real T[18]; /* Cl J */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 6; j++) { real t = 0; for (int k = 0; k < 3; k++) t += Cl[3 * i + k] * J[6 * k + j]; T[6 * i + j] = t; }
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { real t = 0; for (int k = 0; k < 3; k++) t += J[6 * k + i] * T[6 * k + j]; C[6 * i + j] += t; }
}
static void wrench_add(real *f6, const real *p, const real *f, real sign) {
    real m3[3];
    cross3(p, f, m3);
    for (int j = 0; j < 3; j++) { f6[j] += sign * m3[j]; f6[3 + j] += sign * f[j]; }
}
static void vel_at(const real *v6, const real *p, real *out) {
    real t[3];
    cross3(v6, p, t);
    for (int j = 0; j < 3; j++) out[j] = v6[3 + j] + t[j];
}

typedef struct { real R[9], p[3]; const real *v; int ent; } shape_w_t;
static const real ZERO6[6] = {0, 0, 0, 0, 0, 0};

typedef struct { real f[6]; real C[36]; real rep[3]; int any; real wsum; /* sum of the points' ramps (mode 0: patch normalisation) */ } pair_acc_t;

/* one contact point: normal n from B to A, penetration depth > 0 (same law as csrc/mppi_scene.hpp::contact_point,
 * written from DESIGN.md section 3) */
static real g_ramp_depth; /* contact_ramp_depth of the model being stepped (set by scene_contacts; per-thread copies are equal) */
#pragma omp threadprivate(g_ramp_depth)
static void contact_point(int mode, real mu, real k, real cn, real ct, real kh, const real *p, const real *n, real depth,
                          const real *vA, const real *vB, pair_acc_t *acc) {
    /* Hunt-Crossley-style ramp of the velocity-proportional normal terms over the first contact_ramp_depth of penetration:
     * the contact force is continuous at touch-down (include/mppi_hip.h, mppi_model_t.contact_ramp_depth) */
    real ramp = 1;
    /* mode 3: the depth scale of the ramp is 1/MPPI_LIGHT_RAMP_DIV of the static sag (these contacts carry drive forces at a fraction
     * of a millimetre: 20 N at 21 kN/m) - it shapes the stick damper and the patch weights only, see below */
    const real rd = mode == 3 ? g_ramp_depth / (real)MPPI_LIGHT_RAMP_DIV : g_ramp_depth;
    if (rd > 0) { ramp = depth / rd; if (ramp > 1) ramp = 1; }
    real va[3], vb[3], vr[3], vt[3];
    vel_at(vA, p, va); vel_at(vB, p, vb);
    for (int j = 0; j < 3; j++) vr[j] = va[j] - vb[j];
    real vn = vr[0] * n[0] + vr[1] * n[1] + vr[2] * n[2];
    for (int j = 0; j < 3; j++) vt[j] = vr[j] - vn * n[j];
    real vtn = (real)sqrt((double)(vt[0] * vt[0] + vt[1] * vt[1] + vt[2] * vt[2]));
    acc->any = 1;
    acc->wsum += ramp;   /* (patch normalisation of two dynamic bodies, modes 0 and 3) */
    /* the stick cap of the friction viscosity ramps in with the penetration as well: a grazing contact (f_n -> 0) of a body at
     * rest (|v_t| -> 0) would otherwise get the full stick damper c_t from the ratio of two vanishing numbers */
    ct *= ramp;
    if (mode == 0) {
        real fn = k * depth - ramp * cn * vn; if (fn < 0) fn = 0;
        real sc = mu * fn / (vtn + (real)1e-9); if (ct < sc) sc = ct;
        real f[3];
        for (int j = 0; j < 3; j++) { f[j] = fn * n[j] - sc * vt[j]; acc->rep[j] += f[j]; }
        wrench_add(acc->f, p, f, 1);
        return;
    }
    /* Kelvin-Voigt damper on approach AND separation (a law that damps the approach only switches c_n on and off with the
     * sign of v_n and keeps a body that rests on several points rocking at fp32 rounding level), capped so that the force at
     * the start velocity never turns adhesive: k depth - a v_n >= 0 */
    /* (mode 3, a light body held by robot links: damper and end-of-step spring are NOT ramped - a spring that is explicit for the most
     * part throws a 22-gram finger link whose drive has saturated back out of the contact substep after substep, and with the full
     * damper an impact at closing speed loses alpha / (alpha + beta) of its penetration per substep without leaving the contact) */
    real a = mode == 3 ? cn + kh : ramp * (cn + kh);
    if (mode != 3 && vn > 0 && a * vn > k * depth) a = k * depth / vn;
    real fn = k * depth - a * vn; if (fn < 0) fn = 0;
    /* (mode 3: no cap, and the Coulomb limit is taken from the spring alone.  The cap replaces a by k depth / v_n whenever the pair
     * separates faster than the spring alone would push it - which a contact that relaxes by alpha / (alpha + beta) per substep does in
     * EVERY substep after an impact; with the robot's gains on a 22-gram finger that capped damper (h a = tens of kilograms) is honey:
     * the finger keeps its rebound velocity until it has left the contact, its effort drive closes it again at full speed, and the
     * grip chatters with a period of four substeps while f_n = 0 lets the block fall (omnipanda_effort, 6 N per finger).  Uncapped,
     * the implicit solve puts the finger at the velocity where spring, damper and drive balance within one substep; the price is a
     * contact that holds an OPENING finger back for the one or two substeps its last fraction of a millimetre takes to relax.
     * The Coulomb limit is the spring-damper force WITHOUT the implicit spring term, k depth - c_n v_n: in a relaxing contact whose
     * link is pressed on by a force F (v_n+ = (k depth - F) / a, depth shrinking by alpha / (alpha + beta) of its excess per substep)
     * that is exactly F - the 6 N of the finger drive while 3 mm of impact penetration relax, nothing for a block that nothing holds
     * against the link.  k depth - a v_n is zero all through the relaxation (the block falls out of the grip before it is over);
     * k depth alone glues a one-gram block to the side of the stick that has hit it and the stick carries it off the table.) */
    if (mode == 3) { fn = k * depth - cn * vn; if (fn < 0) fn = 0; }
    real b = mu * fn / (vtn + (real)1e-9); if (ct < b) b = ct;
    real f[3];
    for (int j = 0; j < 3; j++) f[j] = k * depth * n[j];
    wrench_add(acc->f, p, f, 1);
    damping_add(acc->C, p, n, a, b);
    for (int j = 0; j < 3; j++) acc->rep[j] += f[j] - (a - b) * vn * n[j] - b * vr[j];
}

static void shape_world(const mppi_model_t *m, const mppi_shape_t *S, int ent, const frame_t *fr, const real *root, shape_w_t *w) {
    real Rf[9], pf[3];
    if (ent >= 0) { memcpy(Rf, fr[ent].R, sizeof Rf); memcpy(pf, fr[ent].p, sizeof pf); w->v = fr[ent].v; }
    else { const real *rs = root + 13 * S->actor; quat_to_R(rs + 3, Rf); pf[0] = rs[0]; pf[1] = rs[1]; pf[2] = rs[2]; w->v = ZERO6; }
    real Rs[9], ps[3], t[3];
    for (int j = 0; j < 9; j++) Rs[j] = (real)S->R[j];
    for (int j = 0; j < 3; j++) ps[j] = (real)S->p[j];
    m3_mul(Rf, Rs, w->R);
    m3_vec(Rf, ps, t);
    for (int j = 0; j < 3; j++) w->p[j] = pf[j] + t[j];
    w->ent = ent;
    (void)m;
}

/* Penetration depth and push-out direction of a point inside a box, continuous everywhere in the interior: with the
 * distances dx, dy, dz > 0 to the three nearest faces, depth = (dx^-2 + dy^-2 + dz^-2)^-1/2 - a smooth minimum that vanishes
 * on every face, equals the nearest-face distance next to a face and blends near edges and corners - and the normal is the
 * unit vector along sum_i max(0, depth/d_i - 1/5)^3 n_i (the gradient's direction with the far faces cut off).  (The nearest-face rule switched the direction of the force by 90
 * degrees where two distances tie; a point leaving through a side face kept its front-face force until the last moment.) */
static void box_interior(real dx, real dy, real dz, const real *y, real *nl, real *depth) {
    real ix = 1 / dx, iy = 1 / dy, iz = 1 / dz;
    real ds = 1 / (real)sqrt((double)(ix * ix + iy * iy + iz * iz));
    /* compact support (round 4): a face further away than five times the depth has NO share in the direction - (depth/d_i)^3 leaked
     * 3e-5 of the side faces' directions into the normal of a block resting 8 mm deep on a 0.5-m table (a creep of 1 um/s) */
    real wx = ds * ix - (real)0.2, wy = ds * iy - (real)0.2, wz = ds * iz - (real)0.2;
    wx = wx > 0 ? wx : 0; wy = wy > 0 ? wy : 0; wz = wz > 0 ? wz : 0;
    wx = wx * wx * wx; wy = wy * wy * wy; wz = wz * wz * wz;
    real nn = 1 / (real)sqrt((double)(wx * wx + wy * wy + wz * wz));
    nl[0] = (y[0] > 0 ? wx : -wx) * nn; nl[1] = (y[1] > 0 ? wy : -wy) * nn; nl[2] = (y[2] > 0 ? wz : -wz) * nn;
    *depth = ds;
}

static void corners_in_box(int mode, real mu, real k, real cn, real ct, real kh, const shape_w_t *X, const double *hx, const shape_w_t *Y,
                           const double *hy, real sign, const real *vA, const real *vB, pair_acc_t *acc) {
    /* feature points of X: 8 corners, 12 edge midpoints, 6 face centres (26 = 3^3 - centre) */
    for (int c = 0; c < 27; c++) {
        if (c == 13) continue;
        real loc[3] = {(real)((c % 3 - 1) * hx[0]), (real)(((c / 3) % 3 - 1) * hx[1]), (real)((c / 9 - 1) * hx[2])};
        real pw[3], t[3], d[3], y[3];
        m3_vec(X->R, loc, t);
        for (int j = 0; j < 3; j++) { pw[j] = X->p[j] + t[j]; d[j] = pw[j] - Y->p[j]; }
        m3_tvec(Y->R, d, y);
        real dx = (real)hy[0] - (real)fabs((double)y[0]), dy = (real)hy[1] - (real)fabs((double)y[1]), dz = (real)hy[2] - (real)fabs((double)y[2]);
        if (dx > 0 && dy > 0 && dz > 0) {
            real nl[3], depth;
            box_interior(dx, dy, dz, y, nl, &depth);
            real n[3];
            m3_vec(Y->R, nl, n);
            for (int j = 0; j < 3; j++) n[j] *= sign;
            contact_point(mode, mu, k, cn, ct, kh, pw, n, depth, vA, vB, acc);
        }
    }
}

/* Two DYNAMIC boxes (mode 0, round 5): ONE normal for the whole pair, from the separating-axis test, instead of a push-out direction per
 * feature point.  The per-point rule (box_interior: towards the nearest face of the OTHER box) fails where it matters for two
 * robots - two equal chassis meeting squarely: every corner and edge midpoint of one lies on a face plane of the other (only
 * the face centres are inside: half the nominal stiffness), and as soon as the boxes pitch a little the points of the top
 * edge are nearer to the other box's TOP face than to its front: they are pushed up, one chassis climbs the other and the two
 * end up inside each other.  Here (csrc/mppi_scene.hpp::box_pair_sat / box_points_along are the same arithmetic):
 *   - 15-axis separating-axis test; an axis that separates: no contact.  depth_sat = the smallest overlap;
 *   - n = blend of the six face axes with weights max(0, 2 - o_a / o_min)^3 - a pure face normal unless two overlaps are
 *     within a factor two of each other -, oriented from B to A;
 *   - every feature point inside the other box is pushed along n; its depth = the distance it has to travel along n to leave
 *     that box (ray exit: continuous in the point and in n);
 *   - a patch that its points under-sample (a lone corner; two edges that cross: NO feature point inside) is filled up to HALF the
 *     nominal stiffness: the share 2 - sum of ramps goes to one more contact of depth_sat at the incident box's support point
 *     (smoothed over +-0.05 in the direction cosine: a face lying flat gives its centre, a tilted one its deepest corner)
 *     clamped onto the reference face.  Half, not all of it: the stability bound of the explicit law (alpha + 2 beta < 4) is
 *     per BODY, and a block held between two fingers sees two patches - filled to the full nominal stiffness the recorded
 *     gripper state lost 0.8 % of its 8192 rollouts to fp32 rounding (99.2 % within 1e-3 of fp64 instead of 99.9 %).  */
typedef struct { int hit; real n[3], p[3], depth; } box_sat_t;
static real sat1(real x) { return x > 1 ? 1 : (x < -1 ? -1 : x); }
static void box_pair_sat(const shape_w_t *A, const double *hA_, const shape_w_t *B, const double *hB_, box_sat_t *out) {
    out->hit = 0;
    real hA[3] = {(real)hA_[0], (real)hA_[1], (real)hA_[2]}, hB[3] = {(real)hB_[0], (real)hB_[1], (real)hB_[2]};
    real C[9], aC[9], d[3], t[3], tA[3];   /* C[3 i + j] = b_i . a_j;  t = centre of A in B's frame;  tA[j] = a_j . (pA - pB) */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        C[3 * i + j] = B->R[i] * A->R[j] + B->R[3 + i] * A->R[3 + j] + B->R[6 + i] * A->R[6 + j];
        aC[3 * i + j] = (real)fabs((double)C[3 * i + j]);
    }
    for (int j = 0; j < 3; j++) d[j] = A->p[j] - B->p[j];
    m3_tvec(B->R, d, t);
    m3_tvec(A->R, d, tA);
    real oB[3], oA[3], omin = (real)1e30, odepth;
    for (int i = 0; i < 3; i++) {
        oB[i] = hB[i] + aC[3 * i] * hA[0] + aC[3 * i + 1] * hA[1] + aC[3 * i + 2] * hA[2] - (real)fabs((double)t[i]);
        oA[i] = hA[i] + aC[i] * hB[0] + aC[3 + i] * hB[1] + aC[6 + i] * hB[2] - (real)fabs((double)tA[i]);
        if (!(oB[i] > 0) || !(oA[i] > 0)) return;
        if (oB[i] < omin) omin = oB[i];
        if (oA[i] < omin) omin = oA[i];
    }
    odepth = omin;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {   /* edge axes b_i x a_j */
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        const real l2 = 1 - C[3 * i + j] * C[3 * i + j];
        if (l2 < (real)1e-3) continue;   /* (nearly parallel edges: the face axes cover this direction) */
        const real o = hB[i1] * aC[3 * i2 + j] + hB[i2] * aC[3 * i1 + j] + hA[j1] * aC[3 * i + j2] + hA[j2] * aC[3 * i + j1]
                       - (real)fabs((double)(t[i2] * C[3 * i1 + j] - t[i1] * C[3 * i2 + j]));
        if (!(o > 0)) return;
        const real on = o / (real)sqrt((double)l2);
        if (on < odepth) odepth = on;
    }
    const real itau = 20;
    real nB[3] = {0, 0, 0}, nA[3] = {0, 0, 0}, yB[3] = {0, 0, 0}, xA[3] = {0, 0, 0}, WB = 0, WA = 0;
    for (int i = 0; i < 3; i++) {   /* B's face i is the reference, A the incident box: everything in B's frame */
        real w = 2 - oB[i] / omin;
        w = w > 0 ? w * w * w : 0;
        const real sg = t[i] > 0 ? (real)1 : (real)-1;
        real y[3] = {t[0], t[1], t[2]};
        for (int j = 0; j < 3; j++) {
            const real sj = hA[j] * sat1(sg * C[3 * i + j] * itau);
            for (int l = 0; l < 3; l++) y[l] -= sj * C[3 * l + j];
        }
        for (int l = 0; l < 3; l++) y[l] = y[l] > hB[l] ? hB[l] : (y[l] < -hB[l] ? -hB[l] : y[l]);
        y[i] = sg * (hB[i] - (real)0.5 * oB[i]);
        nB[i] += w * sg;
        for (int l = 0; l < 3; l++) yB[l] += w * y[l];
        WB += w;
    }
    for (int j = 0; j < 3; j++) {   /* A's face j is the reference, B the incident box: everything in A's frame */
        real w = 2 - oA[j] / omin;
        w = w > 0 ? w * w * w : 0;
        const real sg = tA[j] > 0 ? (real)1 : (real)-1;
        real x[3] = {-tA[0], -tA[1], -tA[2]};   /* centre of B in A's frame */
        for (int i = 0; i < 3; i++) {
            const real si = hB[i] * sat1(sg * C[3 * i + j] * itau);
            for (int l = 0; l < 3; l++) x[l] += si * C[3 * i + l];
        }
        for (int l = 0; l < 3; l++) x[l] = x[l] > hA[l] ? hA[l] : (x[l] < -hA[l] ? -hA[l] : x[l]);
        x[j] = -sg * (hA[j] - (real)0.5 * oA[j]);
        nA[j] += w * sg;
        for (int l = 0; l < 3; l++) xA[l] += w * x[l];
        WA += w;
    }
    real u[3], v[3];
    m3_vec(B->R, nB, u); m3_vec(A->R, nA, v);
    for (int l = 0; l < 3; l++) out->n[l] = u[l] + v[l];
    const real nn2 = out->n[0] * out->n[0] + out->n[1] * out->n[1] + out->n[2] * out->n[2];
    if (!(nn2 > (real)1e-12)) return;   /* (opposite face normals of equal weight cancel: no direction to push in) */
    const real inn = 1 / (real)sqrt((double)nn2);
    for (int l = 0; l < 3; l++) out->n[l] *= inn;
    m3_vec(B->R, yB, u); m3_vec(A->R, xA, v);
    for (int l = 0; l < 3; l++) out->p[l] = (WB * B->p[l] + u[l] + WA * A->p[l] + v[l]) / (WA + WB);
    out->depth = odepth;
    out->hit = 1;
}
/* feature points of X inside Y, all pushed along the pair normal n (world, from B to A); sign = +1 when X is shape A: a point of
 * A leaves B along +n, a point of B leaves A along -n */
static void corners_along(int mode, real mu, real k, real cn, real ct, real kh, const shape_w_t *X, const double *hx, const shape_w_t *Y,
                          const double *hy, real sign, const real *n, const real *vA, const real *vB, pair_acc_t *acc) {
    real nl[3], inv[3];
    m3_tvec(Y->R, n, nl);
    for (int l = 0; l < 3; l++) { nl[l] *= sign; const real a = (real)fabs((double)nl[l]); inv[l] = 1 / (a > (real)1e-9 ? a : (real)1e-9); }   /* (a face the ray runs parallel to is never its exit) */
    for (int c = 0; c < 27; c++) {
        if (c == 13) continue;
        real loc[3] = {(real)((c % 3 - 1) * hx[0]), (real)(((c / 3) % 3 - 1) * hx[1]), (real)((c / 9 - 1) * hx[2])};
        real pw[3], t[3], d[3], y[3];
        m3_vec(X->R, loc, t);
        for (int j = 0; j < 3; j++) { pw[j] = X->p[j] + t[j]; d[j] = pw[j] - Y->p[j]; }
        m3_tvec(Y->R, d, y);
        real dx = (real)hy[0] - (real)fabs((double)y[0]), dy = (real)hy[1] - (real)fabs((double)y[1]), dz = (real)hy[2] - (real)fabs((double)y[2]);
        if (dx > 0 && dy > 0 && dz > 0) {
            real depth = (real)1e30;
            /* ray exit from Y along nl, capped at four times the distance to the NEAREST face: a point that enters through a side
             * face (a finger sliding over the block) starts at depth 0 and gains four times its distance from that face until the
             * exit along n takes over - with the plain ray exit its depth jumped to the pair's penetration the moment it was inside */
            const real near = dx < dy ? (dx < dz ? dx : dz) : (dy < dz ? dy : dz);
            depth = 4 * near;
            for (int l = 0; l < 3; l++) {
                const real e = ((real)hy[l] - (nl[l] > 0 ? y[l] : -y[l])) * inv[l];
                if (e < depth) depth = e;
            }
            contact_point(mode, mu, k, cn, ct, kh, pw, n, depth, vA, vB, acc);
        }
    }
}

/* sphere (centre ps, radius r) against box Y; sign = +1 when the sphere is shape A */
static void sphere_in_box(int mode, real mu, real k, real cn, real ct, real kh, const real *ps, real r, const shape_w_t *Y, const double *hy,
                          real sign, const real *vA, const real *vB, pair_acc_t *acc) {
    real d[3], y[3], cl[3], e[3], nl[3] = {0, 0, 0}, depth, dist2 = 0;
    for (int j = 0; j < 3; j++) d[j] = ps[j] - Y->p[j];
    m3_tvec(Y->R, d, y);
    for (int j = 0; j < 3; j++) {
        cl[j] = y[j] < -(real)hy[j] ? -(real)hy[j] : (y[j] > (real)hy[j] ? (real)hy[j] : y[j]);
        e[j] = y[j] - cl[j];
        dist2 += e[j] * e[j];
    }
    if (dist2 >= r * r) return;
    if (dist2 > (real)1e-12) {
        real dist = (real)sqrt((double)dist2);
        for (int j = 0; j < 3; j++) nl[j] = e[j] / dist;
        depth = r - dist;
    } else {
        real dx = (real)hy[0] - (real)fabs((double)y[0]), dy = (real)hy[1] - (real)fabs((double)y[1]), dz = (real)hy[2] - (real)fabs((double)y[2]);
        real tiny = (real)1e-9;  /* centre on the surface: the smooth minimum needs positive distances */
        box_interior(dx > tiny ? dx : tiny, dy > tiny ? dy : tiny, dz > tiny ? dz : tiny, y, nl, &depth);
        depth += r;
    }
    real pw[3], n[3], t[3];
    m3_vec(Y->R, cl, t);
    for (int j = 0; j < 3; j++) pw[j] = Y->p[j] + t[j];
    m3_vec(Y->R, nl, n);
    for (int j = 0; j < 3; j++) n[j] *= sign;
    contact_point(mode, mu, k, cn, ct, kh, pw, n, depth, vA, vB, acc);
}

/* disc (wheel / caster: centre pc, unit axis ax, radius r) against box Y - round 5, the reference's "everything in an env
 * collides" (isaacgym_wrapper.py:436-442) for the wheels and casters of the mobile bases.  One analytic point, the
 * deepest point of the disc in the box: e = direction from the disc centre into the box (to the box's closest point; with the
 * centre inside: against box_interior's push-out normal), e_p its part in the disc's plane, contact point = centre + r e_p / |e|
 * (rim point for a box in the disc's plane, centre for a box over the flat side); a box thinner than the disc reaches along that
 * ray is met in the middle of its stretch (slab test).  Depth and normal of a point inside a box. */
static void disc_in_box(int mode, real mu, real k, real cn, real ct, real kh, const real *pc, const real *ax, real r, const shape_w_t *Y, const double *hy,
                        real sign, const real *vA, const real *vB, pair_acc_t *acc) {
    real d[3], yc[3], al[3], e[3], ep[3], y[3], c3[3];
    for (int j = 0; j < 3; j++) d[j] = pc[j] - Y->p[j];
    m3_tvec(Y->R, d, yc);
    m3_tvec(Y->R, ax, al);
    int inside = 1;
    for (int j = 0; j < 3; j++) {
        real cl = yc[j] < -(real)hy[j] ? -(real)hy[j] : (yc[j] > (real)hy[j] ? (real)hy[j] : yc[j]);
        e[j] = cl - yc[j];
        c3[j] = (real)hy[j] - (real)fabs((double)yc[j]);
        if (!(c3[j] > 0)) inside = 0;
    }
    if (inside) {
        real nc[3], dc;
        box_interior(c3[0], c3[1], c3[2], yc, nc, &dc);
        for (int j = 0; j < 3; j++) e[j] = -nc[j];
    }
    real le2 = e[0] * e[0] + e[1] * e[1] + e[2] * e[2];
    if (!(le2 > (real)1e-20)) return;
    real ea = e[0] * al[0] + e[1] * al[1] + e[2] * al[2], lp2 = 0;
    for (int j = 0; j < 3; j++) { ep[j] = e[j] - ea * al[j]; lp2 += ep[j] * ep[j]; }
    real t = r * (real)sqrt((double)(lp2 / le2));
    for (int j = 0; j < 3; j++) y[j] = yc[j];
    if (lp2 > (real)1e-12 * le2) {
        real il = 1 / (real)sqrt((double)lp2), t_in = (real)-1e30, t_out = (real)1e30;
        for (int j = 0; j < 3; j++) {
            real u = ep[j] * il;
            if (fabs((double)u) > 1e-6) {
                real t1 = (-(real)hy[j] - yc[j]) / u, t2 = ((real)hy[j] - yc[j]) / u;
                real lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
                if (lo > t_in) t_in = lo;
                if (hi < t_out) t_out = hi;
            }
        }
        if (t_out < (real)1e29 && t_in > (real)-1e29) {
            real mid = (real)0.5 * ((t_in > 0 ? t_in : 0) + t_out);
            if (mid < t) t = mid;
        }
        for (int j = 0; j < 3; j++) y[j] = yc[j] + t * ep[j] * il;
    }
    real dx = (real)hy[0] - (real)fabs((double)y[0]), dy = (real)hy[1] - (real)fabs((double)y[1]), dz = (real)hy[2] - (real)fabs((double)y[2]);
    if (!(dx > 0 && dy > 0 && dz > 0)) return;
    real nl[3], depth, pw[3], n[3], t3[3];
    box_interior(dx, dy, dz, y, nl, &depth);
    m3_vec(Y->R, y, t3);
    for (int j = 0; j < 3; j++) pw[j] = Y->p[j] + t3[j];
    m3_vec(Y->R, nl, n);
    for (int j = 0; j < 3; j++) n[j] *= sign;
    contact_point(mode, mu, k, cn, ct, kh, pw, n, depth, vA, vB, acc);
}

/* disc against a sphere (round 5: wheels against the sphere obstacles of the benchmark adapters): the disc's point nearest to the
 * sphere centre (its projection into the disc's plane, pulled back onto the disc) against the sphere's surface */
static void disc_sphere(int mode, real mu, real k, real cn, real ct, real kh, const real *pc, const real *ax, real r, const real *ps, real rs,
                        real sign, const real *vA, const real *vB, pair_acc_t *acc) {
    real e[3], q[3], d[3], n[3], ea = 0, l2 = 0, d2 = 0;
    for (int j = 0; j < 3; j++) { e[j] = ps[j] - pc[j]; ea += e[j] * ax[j]; }
    for (int j = 0; j < 3; j++) { e[j] -= ea * ax[j]; l2 += e[j] * e[j]; }
    real sc = l2 > r * r ? r / (real)sqrt((double)l2) : 1;
    for (int j = 0; j < 3; j++) { q[j] = pc[j] + sc * e[j]; d[j] = q[j] - ps[j]; d2 += d[j] * d[j]; }
    if (d2 >= rs * rs) return;
    real dist = 0;
    for (int j = 0; j < 3; j++) n[j] = ax[j];
    if (d2 > (real)1e-12) { dist = (real)sqrt((double)d2); for (int j = 0; j < 3; j++) n[j] = d[j] / dist; }
    for (int j = 0; j < 3; j++) n[j] *= sign;
    contact_point(mode, mu, k, cn, ct, kh, q, n, rs - dist, vA, vB, acc);
}

/* two spheres: normal along the line of centres (from B to A), contact point in the middle of the overlap; coincident centres
 * push apart along +z.  (The plannerbenchmark adapters add sphere obstacles next to sphere-shaped robot links,
 * benchmarks/panda_arm/mppi_planner/mppi_planner_wrapper.py:58-79.) */
static void sphere_sphere(int mode, real mu, real k, real cn, real ct, real kh, const real *pa, real ra, const real *pb, real rb,
                          const real *vA, const real *vB, pair_acc_t *acc) {
    real d[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
    real dist2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], rs = ra + rb;
    if (dist2 >= rs * rs) return;
    real n[3] = {0, 0, 1}, dist = 0;
    if (dist2 > (real)1e-12) { dist = (real)sqrt((double)dist2); for (int j = 0; j < 3; j++) n[j] = d[j] / dist; }
    real depth = rs - dist, pw[3];
    for (int j = 0; j < 3; j++) pw[j] = pb[j] + n[j] * (rb - (real)0.5 * depth);
    contact_point(mode, mu, k, cn, ct, kh, pw, n, depth, vA, vB, acc);
}

/* dynamic frame index of a shape (-1: static), and the mass that scales its contact gains */
static int shape_entity(const mppi_model_t *m, const scene_info_t *si, const mppi_shape_t *S, real *mass) {
    *mass = -1;
    if (S->actor == m->robot_actor) {
        if (S->body >= 0) { *mass = si->robot_mass[base_of_body(m, S->body)]; return S->body; }
        if (si->floating) { *mass = si->robot_mass[-1 - S->body]; return m->n_bodies + (-1 - S->body); }
        return -1;
    }
    for (int f = 0; f < si->n_free; f++)
        if (si->free_actor[f] == S->actor) { *mass = (real)m->actors[S->actor].mass; return m->n_bodies + si->nb + f; }
    return -1;
}

/* LIGHT bodies (round 6; the 1-gram block of the reference's examples/panda_pick between the fingers of a 17-kg arm,
 * conf/actors/panda_pick_block.yaml - PhysX's implicit solver holds and lifts it, isaacgym_wrapper.py:29-36).  The explicit law of two
 * dynamic bodies is as stiff as the LIGHTER body can carry in an explicit step (1.3 N/m for one gram at h = 25 ms - a finger drive
 * closes the fingers THROUGH the block) and its stick damper creeps at g h.  A free actor of at most MPPI_LIGHT_BODY_MASS that the robot
 * outweighs MPPI_LIGHT_BODY_RATIO times (heavier bodies carry pushing forces within millimetres under the explicit law, and keep it) and
 * that touches no other free actor - scene_info_t.light; MPPI_CONTACT_EXPLICIT_LIGHT switches it off - takes, against a robot link X
 * (mode 3; geometry and patch normalisation as for two dynamic bodies):
 *   - the ROBOT's gains (k = alpha m_robot / h^2 ...) and the implicit point law with damper and end-of-step spring NOT ramped and the
 *     damper NOT capped at k depth / v_n, the Coulomb limit taken from the spring alone
 *     (contact_point, law 3: the ramp - over 1 / MPPI_LIGHT_RAMP_DIV of the static sag - shapes the stick damper and the patch weights);
 *   - implicit on BOTH bodies, staggered.  With c = J^T (b 1 + (a - b) n n^T) J summed over the pair's points and f its spring wrench on L,
 *         link X :  wrench = -f - c (v_X+ - v_L)     c joins the link's articulated inertia like a static contact's, the light body is a
 *                                                    wall that moves with its velocity at the START of the substep;
 *         body L :  wrench = +f - c (v_L+ - v_X+)    solved AFTER the robot:  f + c v_X - c v_L+  at the contact pass, and the links'
 *                                                    velocity CHANGES over the substep, dv_X = v_X+ - v_X, as
 *                                                        (sum_X c_X) dv_ref + sum_X (c_X S_X) dqd_X
 *                                                    ref = the frame nearest the base among the PARENTS of the links in contact, S_X /
 *                                                    dqd_X the link's own joint axis / rate change: exact when the links in contact
 *                                                    hang off one parent (two fingers on a hand) - their relative motion is what a
 *                                                    pinch is made of -, and one substep late only for joints between `ref` and a
 *                                                    link's parent (hand AND fingers in contact: the wrist roll under the fingers).
 *                                                    light_pair_t carries sum c, ref and up to LIGHT_RECORDS vectors c_X S_X.
 *     (Tried and dropped: every link's change through an effective contact POINT - a force where the patch has a damping matrix:
 *     what falls into a direction the patch damps weakly spins a gram to 1000 rad/s; all changes through one common reference
 *     link - two fingers both a substep late: the pinch rings at the substep rate for ever.)
 *     Both solves are unconditionally stable (a squeeze between two fingers contracts by m / (m + h c) per substep); the light body
 *     follows the links that hold it without a substep of lag; what the robot feels of it is its weight and an added mass h c while
 *     it accelerates.
 * Its contacts with STATIC geometry (table, ground) keep the law of modes 1 / 2 with the gains of its own mass: at rest it lies still to
 * 1e-7 m/s in the sag of that law; with the robot's gains and no ramp the equilibrium depth of a gram is half a micrometre and the
 * nine feature points of a face toggle in and out of contact for ever (tried: a rocking of 0.3 rad/s that never dies).  The price: a
 * finger can press the block INTO the table until the finger itself meets it. */
#define LIGHT_RECORDS 4
typedef struct {   /* per light body */
    int light, ref;                  /* its frame; the reference frame (below) */
    real C[36];                      /* sum of c over its pairs with robot links */
    int n_rec, joint[LIGHT_RECORDS]; /* links in contact: their own joints ... */
    real g[LIGHT_RECORDS][6];        /* ... with (sum of the link's c) S_joint */
} light_pair_t;

static void scene_contacts(const mppi_model_t *m, const scene_info_t *si, frame_t *fr, const real *root, real *cf, light_pair_t *lp, int *n_lp) {
    real h = (real)(m->dt / m->substeps);
    *n_lp = 0;
    int nf = m->n_bodies + si->nb + MPPI_MAX_FREE;
    g_ramp_depth = (real)m->contact_ramp_depth;
    for (int e = 0; e < nf; e++) { memset(fr[e].f, 0, sizeof fr[e].f); memset(fr[e].C, 0, sizeof fr[e].C); }
    for (int j = 0; j < 3 * m->n_rb; j++) cf[j] = 0;
    for (int ip = 0; ip < m->n_pairs; ip++) {
        const mppi_shape_t *A = &m->shapes[m->pairs[ip].a];
        const mppi_shape_t *B = m->pairs[ip].b >= 0 ? &m->shapes[m->pairs[ip].b] : NULL;
        real ma, mb = -1;
        int ea = shape_entity(m, si, A, &ma), eb = -1;
        if (B) eb = shape_entity(m, si, B, &mb);
        real mua = (real)A->friction, mub = B ? (real)B->friction : (real)m->ground_friction;
        real mu = mua < mub ? mua : mub;
        int mode; real meff;
        if (ma > 0 && mb > 0) { mode = 0; meff = ma * mb / (ma + mb); }
        else if (ma > 0) { mode = 1; meff = ma; }
        else { mode = 2; meff = mb; }
        /* a light body's pair: the robot's gains and law 3; against a robot link: mode 3 (implicit on both bodies) */
        const int la = si->light[A->actor], lb = B ? si->light[B->actor] : 0;
        int law = mode;
        if ((la || lb) && mode == 0) { mode = law = 3; meff = la ? mb : ma; }
        int tb = B ? B->type : -1;
        real npts = (A->type == MPPI_SHAPE_BOX && (tb == MPPI_SHAPE_BOX || tb == -1)) ? 4 : 1;
        /* two dynamic bodies (explicit law): a fixed 1/npts share per point lets a face-to-face contact of 18 feature points
         * carry 4.5 times the nominal stiffness - beyond the stability limit of an explicit spring-damper at this step for the
         * lighter body (alpha + 2 beta < 4 per unit of n / npts).  Their points carry the FULL gains and the pair's summed force
         * is divided by max(npts, sum of the points' ramps): the stiffness of the patch never exceeds the nominal one, and it is
         * continuous in the number of points that take part */
        const real npts_nom = npts;
        const int dyn = mode == 0 || mode == 3;   /* two dynamic bodies: patch-normalised, one normal per pair of boxes */
        if (dyn) npts = 1;
        real k = (real)m->contact_alpha * meff / (h * h) / npts, cn = (real)m->contact_beta * meff / h / npts;
        real ct = (real)m->friction_beta * meff / h / npts, kh = (real)m->contact_alpha * meff / h / npts;
        shape_w_t wa, wb;
        shape_world(m, A, ea, fr, root, &wa);
        pair_acc_t acc;
        memset(&acc, 0, sizeof acc);
        if (!B) {
            real ez[3] = {0, 0, 1};
            if (A->type == MPPI_SHAPE_BOX) {
                for (int c = 0; c < 8; c++) {
                    real loc[3] = {(real)((c & 1) ? A->size[0] : -A->size[0]), (real)((c & 2) ? A->size[1] : -A->size[1]), (real)((c & 4) ? A->size[2] : -A->size[2])};
                    real t[3], pw[3];
                    m3_vec(wa.R, loc, t);
                    for (int j = 0; j < 3; j++) pw[j] = wa.p[j] + t[j];
                    if (pw[2] < 0) contact_point(law, mu, k, cn, ct, kh, pw, ez, -pw[2], wa.v, ZERO6, &acc);
                }
            } else if (A->type == MPPI_SHAPE_SPHERE) {
                real pw[3] = {wa.p[0], wa.p[1], wa.p[2] - (real)A->size[0]};
                if (pw[2] < 0) contact_point(law, mu, k, cn, ct, kh, pw, ez, -pw[2], wa.v, ZERO6, &acc);
            } else { /* disc: lowest rim point; axis = local z of the shape frame */
                real ax[3] = {wa.R[2], wa.R[5], wa.R[8]};
                real d[3] = {ax[2] * ax[0], ax[2] * ax[1], ax[2] * ax[2] - 1};
                real l2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
                if (l2 > (real)1e-8) {
                    real sc = (real)A->size[0] / (real)sqrt((double)l2), pw[3];
                    for (int j = 0; j < 3; j++) pw[j] = wa.p[j] + sc * d[j];
                    if (pw[2] < 0) contact_point(law, mu, k, cn, ct, kh, pw, ez, -pw[2], wa.v, ZERO6, &acc);
                }
            }
        } else {
            shape_world(m, B, eb, fr, root, &wb);
            if (A->type == MPPI_SHAPE_BOX && B->type == MPPI_SHAPE_BOX) {
                if (dyn && !(m->contact_flags & MPPI_CONTACT_POINT_NORMALS)) {
                    box_sat_t sat;
                    box_pair_sat(&wa, A->size, &wb, B->size, &sat);
                    if (sat.hit) {
                        corners_along(law, mu, k, cn, ct, kh, &wa, A->size, &wb, B->size, 1, sat.n, wa.v, wb.v, &acc);
                        corners_along(law, mu, k, cn, ct, kh, &wb, B->size, &wa, A->size, -1, sat.n, wa.v, wb.v, &acc);
                        const real deficit = (real)0.5 * npts_nom - acc.wsum;   /* (HALF the nominal patch, see above) */
                        if (deficit > 0) {
                            pair_acc_t one;
                            memset(&one, 0, sizeof one);
                            contact_point(law, mu, k, cn, ct, kh, sat.p, sat.n, sat.depth, wa.v, wb.v, &one);
                            for (int l = 0; l < 6; l++) acc.f[l] += deficit * one.f[l];
                            for (int l = 0; l < 36; l++) acc.C[l] += deficit * one.C[l];   /* (mode 3; zero in mode 0) */
                            for (int l = 0; l < 3; l++) acc.rep[l] += deficit * one.rep[l];
                            acc.wsum += deficit * one.wsum;
                            acc.any = 1;
                        }
                    }
                } else {
                    corners_in_box(law, mu, k, cn, ct, kh, &wa, A->size, &wb, B->size, 1, wa.v, wb.v, &acc);
                    corners_in_box(law, mu, k, cn, ct, kh, &wb, B->size, &wa, A->size, -1, wa.v, wb.v, &acc);
                }
            } else if (A->type == MPPI_SHAPE_SPHERE && B->type == MPPI_SHAPE_BOX) {
                sphere_in_box(law, mu, k, cn, ct, kh, wa.p, (real)A->size[0], &wb, B->size, 1, wa.v, wb.v, &acc);
            } else if (A->type == MPPI_SHAPE_BOX && B->type == MPPI_SHAPE_SPHERE) {
                sphere_in_box(law, mu, k, cn, ct, kh, wb.p, (real)B->size[0], &wa, A->size, -1, wa.v, wb.v, &acc);
            } else if (A->type == MPPI_SHAPE_SPHERE && B->type == MPPI_SHAPE_SPHERE) {
                sphere_sphere(law, mu, k, cn, ct, kh, wa.p, (real)A->size[0], wb.p, (real)B->size[0], wa.v, wb.v, &acc);
            } else if (A->type == MPPI_SHAPE_DISC && B->type == MPPI_SHAPE_BOX) {
                real ax[3] = {wa.R[2], wa.R[5], wa.R[8]};
                disc_in_box(law, mu, k, cn, ct, kh, wa.p, ax, (real)A->size[0], &wb, B->size, 1, wa.v, wb.v, &acc);
            } else if (A->type == MPPI_SHAPE_BOX && B->type == MPPI_SHAPE_DISC) {
                real ax[3] = {wb.R[2], wb.R[5], wb.R[8]};
                disc_in_box(law, mu, k, cn, ct, kh, wb.p, ax, (real)B->size[0], &wa, A->size, -1, wa.v, wb.v, &acc);
            } else if (A->type == MPPI_SHAPE_DISC && B->type == MPPI_SHAPE_SPHERE) {
                real ax[3] = {wa.R[2], wa.R[5], wa.R[8]};
                disc_sphere(law, mu, k, cn, ct, kh, wa.p, ax, (real)A->size[0], wb.p, (real)B->size[0], 1, wa.v, wb.v, &acc);
            } else if (A->type == MPPI_SHAPE_SPHERE && B->type == MPPI_SHAPE_DISC) {
                real ax[3] = {wb.R[2], wb.R[5], wb.R[8]};
                disc_sphere(law, mu, k, cn, ct, kh, wb.p, ax, (real)B->size[0], wa.p, (real)A->size[0], -1, wa.v, wb.v, &acc);
            }
        }
        if (!acc.any) continue;
        if (dyn) {
            const real sc = 1 / (acc.wsum > npts_nom ? acc.wsum : npts_nom);
            for (int j = 0; j < 6; j++) acc.f[j] *= sc;
            for (int j = 0; j < 36; j++) acc.C[j] *= sc;
            for (int j = 0; j < 3; j++) acc.rep[j] *= sc;
        }
        if (mode == 0) {
            for (int j = 0; j < 6; j++) { fr[ea].f[j] += acc.f[j]; fr[eb].f[j] -= acc.f[j]; }
        } else if (mode == 3) {
            const int heavy = la ? eb : ea, light = la ? ea : eb;
            real CvL[6], CvX[6];
            m6_vec(acc.C, fr[light].v, CvL);           /* the light body as a moving wall: + c v_L(start) on the link */
            m6_vec(acc.C, fr[heavy].v, CvX);           /* ... and the link's start velocity on the light body */
            for (int j = 0; j < 6; j++) { fr[ea].f[j] += acc.f[j]; fr[eb].f[j] -= acc.f[j]; fr[heavy].f[j] += CvL[j]; fr[light].f[j] += CvX[j]; }
            for (int j = 0; j < 36; j++) { fr[ea].C[j] += acc.C[j]; fr[eb].C[j] += acc.C[j]; }
            light_pair_t *r = NULL;                   /* the light body's record */
            const int n = m->n_bodies;
            for (int l = 0; l < *n_lp; l++) if (lp[l].light == light) r = &lp[l];
            if (!r) { r = &lp[(*n_lp)++]; memset(r, 0, sizeof *r); r->light = light; r->ref = -1; }
            for (int j = 0; j < 36; j++) r->C[j] += acc.C[j];
            /* parent frame of the link (a shape on a moving base: the base itself), nearest the base wins */
            const int par = heavy >= n ? heavy : (m->bodies[heavy].parent < 0 ? n + (-1 - m->bodies[heavy].parent) : m->bodies[heavy].parent);
            if (r->ref < 0 || (par >= n && r->ref < n) || (par < n && r->ref < n && par < r->ref)) r->ref = par;
            if (heavy < n) {
                int k = 0;
                while (k < r->n_rec && r->joint[k] != heavy) k++;
                if (k < LIGHT_RECORDS) {
                    if (k == r->n_rec) { r->joint[k] = heavy; r->n_rec++; }
                    const mppi_body_t *b = &m->bodies[heavy];
                    real ax[3] = {(real)b->axis[0], (real)b->axis[1], (real)b->axis[2]}, aw[3], t[3], Sw[6], g6[6];
                    m3_vec(fr[heavy].R, ax, aw);
                    if (b->jtype == MPPI_JOINT_REVOLUTE) { cross3(fr[heavy].p, aw, t); for (int j = 0; j < 3; j++) { Sw[j] = aw[j]; Sw[3 + j] = t[j]; } }
                    else for (int j = 0; j < 3; j++) { Sw[j] = 0; Sw[3 + j] = aw[j]; }
                    m6_vec(acc.C, Sw, g6);
                    for (int j = 0; j < 6; j++) r->g[k][j] += g6[j];
                }
            }
        } else if (mode == 1) {
            for (int j = 0; j < 6; j++) fr[ea].f[j] += acc.f[j];
            for (int j = 0; j < 36; j++) fr[ea].C[j] += acc.C[j];
        } else {
            for (int j = 0; j < 6; j++) fr[eb].f[j] -= acc.f[j];
            for (int j = 0; j < 36; j++) fr[eb].C[j] += acc.C[j];
        }
        for (int j = 0; j < 3; j++) { cf[3 * A->rb + j] += acc.rep[j]; if (B) cf[3 * B->rb + j] -= acc.rep[j]; }
    }
}

/* rigid inertia about the world origin, world axes, from body-frame (m, h = m c, Io about the body origin) */
static void world_inertia(const real *R, const real *p, real mass, const double *h_b, const double *Io6, real *I6, real *hw) {
    real c[3] = {0, 0, 0}, hb[3] = {(real)h_b[0], (real)h_b[1], (real)h_b[2]};
    if (mass > 0) for (int j = 0; j < 3; j++) c[j] = hb[j] / mass;
    real cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    real Ic[9];
    real Io[9] = {(real)Io6[0], (real)Io6[1], (real)Io6[2], (real)Io6[1], (real)Io6[3], (real)Io6[4], (real)Io6[2], (real)Io6[4], (real)Io6[5]};
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ic[3 * i + j] = Io[3 * i + j] - mass * ((i == j ? cc : 0) - c[i] * c[j]);
    real T[9], Rt[9], Iw[9], cw[3], t[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = R[3 * j + i];
    m3_mul(R, Ic, T); m3_mul(T, Rt, Iw);
    m3_vec(R, c, t);
    for (int j = 0; j < 3; j++) { cw[j] = p[j] + t[j]; hw[j] = mass * cw[j]; }
    real cw2 = cw[0] * cw[0] + cw[1] * cw[1] + cw[2] * cw[2];
    real Iow[6] = {Iw[0] + mass * (cw2 - cw[0] * cw[0]), Iw[1] - mass * cw[0] * cw[1], Iw[2] - mass * cw[0] * cw[2],
                   Iw[4] + mass * (cw2 - cw[1] * cw[1]), Iw[5] - mass * cw[1] * cw[2], Iw[8] + mass * (cw2 - cw[2] * cw[2])};
    rigid_inertia(mass, hw, Iow, I6);
}

static int solve6(real *A, real *b) { /* Gaussian elimination with partial pivoting, in place */
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++) if (fabs((double)A[6 * r + c]) > fabs((double)A[6 * piv + c])) piv = r;
        if (A[6 * piv + c] == 0) return -1;
        if (piv != c) { for (int j = 0; j < 6; j++) { real t = A[6 * c + j]; A[6 * c + j] = A[6 * piv + j]; A[6 * piv + j] = t; } real t = b[c]; b[c] = b[piv]; b[piv] = t; }
        for (int r = c + 1; r < 6; r++) {
            real f = A[6 * r + c] / A[6 * c + c];
            for (int j = c; j < 6; j++) A[6 * r + j] -= f * A[6 * c + j];
            b[r] -= f * b[c];
        }
    }
    for (int r = 5; r >= 0; r--) { real t = b[r]; for (int j = r + 1; j < 6; j++) t -= A[6 * r + j] * b[j]; b[r] = t / A[6 * r + r]; }
    return 0;
}

static void quat_integrate(real *q, const real *w, real h) {
    real x = q[0], y = q[1], z = q[2], s = q[3], k = h / 2;
    real nx = x + k * (w[0] * s + w[1] * z - w[2] * y), ny = y + k * (w[1] * s + w[2] * x - w[0] * z);
    real nz = z + k * (w[2] * s + w[0] * y - w[1] * x), ns = s - k * (w[0] * x + w[1] * y + w[2] * z);
    real n = (real)sqrt((double)(nx * nx + ny * ny + nz * nz + ns * ns));
    q[0] = nx / n; q[1] = ny / n; q[2] = nz / n; q[3] = ns / n;
}
static void root_integrate(real *rs, const real *a6, real h) {
    real *p = rs, *v = rs + 7, *w = rs + 10, t1[3], t2[3], vd[3];
    cross3(a6, p, t1); cross3(w, v, t2);
    for (int j = 0; j < 3; j++) vd[j] = a6[3 + j] + t1[j] + t2[j];
    for (int j = 0; j < 3; j++) w[j] += h * a6[j];
    for (int j = 0; j < 3; j++) v[j] += h * vd[j];
    for (int j = 0; j < 3; j++) p[j] += h * v[j];
    quat_integrate(rs + 3, w, h);
}
static void root_frame(const real *rs, frame_t *f) {
    quat_to_R(rs + 3, f->R);
    real t[3];
    cross3(rs + 10, rs, t);
    for (int j = 0; j < 3; j++) { f->p[j] = rs[j]; f->v[j] = rs[10 + j]; f->v[3 + j] = rs[7 + j] - t[j]; }
}

/* World-frame dense ABA of the robot with external wrenches / implicit dampings per frame. */
static void scene_aba(const mppi_model_t *m, const scene_info_t *si, frame_t *fr, const real *qd, const real *tau_exp, const real *kdh,
                      real *qdd, real (*abase)[6]) {
    int n = m->n_bodies, nb = si->nb;
    real h = (real)(m->dt / m->substeps);
    real gr[NBASEMAX][3];   /* gravity per tree (ActorWrapper.gravity of the robot that owns it) */
    for (int r = 0; r < nb; r++)
        for (int j = 0; j < 3; j++) gr[r][j] = m->actors[base_actor(m, r)].gravity ? (real)m->gravity[j] : 0;
    real S[NBMAX][6], c[NBMAX][6], IA[NBMAX + NBASEMAX][36], pA[NBMAX + NBASEMAX][6], U[NBMAX][6], d[NBMAX], u[NBMAX], a[NBMAX][6];
    for (int i = 0; i < n + nb; i++) {
        const frame_t *F = &fr[i];
        const int r = i < n ? base_of_body(m, i) : i - n;
        const real *g = gr[r];
        real hw[3];
        if (i < n) world_inertia(F->R, F->p, (real)m->bodies[i].mass, m->bodies[i].h, m->bodies[i].Io, IA[i], hw);
        else world_inertia(F->R, F->p, base_mass_of(m, r), base_h_of(m, r), base_Io_of(m, r), IA[i], hw);
        real mass = i < n ? (real)m->bodies[i].mass : base_mass_of(m, r);
        real Iv[6], Cv[6], fg[6], t3[3];
        m6_vec(IA[i], F->v, Iv);
        crf(F->v, Iv, pA[i]);
        m6_vec(F->C, F->v, Cv);
        cross3(hw, g, t3);
        for (int j = 0; j < 3; j++) { fg[j] = t3[j]; fg[3 + j] = mass * g[j]; }
        for (int j = 0; j < 6; j++) pA[i][j] += Cv[j] - F->f[j] - fg[j];
        for (int j = 0; j < 36; j++) IA[i][j] += h * F->C[j];
        if (i < n) {
            const mppi_body_t *b = &m->bodies[i];
            real ax[3] = {(real)b->axis[0], (real)b->axis[1], (real)b->axis[2]}, aw[3], t[3];
            m3_vec(F->R, ax, aw);
            if (b->jtype == MPPI_JOINT_REVOLUTE) { cross3(F->p, aw, t); for (int j = 0; j < 3; j++) { S[i][j] = aw[j]; S[i][3 + j] = t[j]; } }
            else for (int j = 0; j < 3; j++) { S[i][j] = 0; S[i][3 + j] = aw[j]; }
            const real *vp = b->parent < 0 ? fr[n + (-1 - b->parent)].v : fr[b->parent].v;
            real sj[6];
            for (int j = 0; j < 6; j++) sj[j] = S[i][j] * qd[i];
            crm(vp, sj, c[i]);
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        m6_vec(IA[i], S[i], U[i]);
        real sd = 0, sp = 0;
        for (int j = 0; j < 6; j++) { sd += S[i][j] * U[i][j]; sp += S[i][j] * pA[i][j]; }
        d[i] = sd + kdh[i];
        u[i] = tau_exp[i] - sp;
        int par = m->bodies[i].parent < 0 ? n + (-1 - m->bodies[i].parent) : m->bodies[i].parent;
        real Ia[36], t6[6];
        for (int r = 0; r < 6; r++) for (int cc = 0; cc < 6; cc++) Ia[6 * r + cc] = IA[i][6 * r + cc] - U[i][r] * U[i][cc] / d[i];
        m6_vec(Ia, c[i], t6);
        for (int j = 0; j < 6; j++) pA[par][j] += pA[i][j] + t6[j] + U[i][j] * (u[i] / d[i]);
        for (int j = 0; j < 36; j++) IA[par][j] += Ia[j];
    }
    for (int r = 0; r < nb; r++) {   /* one 6x6 base system per tree */
        for (int j = 0; j < 6; j++) abase[r][j] = 0;
        if (si->floating) {
            real A[36], b6[6];
            memcpy(A, IA[n + r], sizeof A);
            for (int j = 0; j < 6; j++) b6[j] = -pA[n + r][j];
            solve6(A, b6);
            memcpy(abase[r], b6, sizeof b6);
        }
    }
    for (int i = 0; i < n; i++) {
        int par = m->bodies[i].parent;
        real ap[6], ua = 0;
        for (int j = 0; j < 6; j++) ap[j] = (par < 0 ? abase[-1 - par][j] : a[par][j]) + c[i][j];
        for (int j = 0; j < 6; j++) ua += U[i][j] * ap[j];
        qdd[i] = (u[i] - ua) / d[i];
        for (int j = 0; j < 6; j++) a[i][j] = ap[j] + S[i][j] * qdd[i];
    }
}

/* frames of the current scene state: robot bodies (world poses + velocities), base, free actors */
static void scene_frames(const mppi_model_t *m, const scene_info_t *si, const real *root, const real *q, const real *qd, frame_t *fr) {
    int n = m->n_bodies;
    kin_t k;
    kinematics(m, root, q, qd, &k);  /* poses only are taken from the body-frame kinematics */
    for (int r = 0; r < si->nb; r++) {
        root_frame(root + 13 * base_actor(m, r), &fr[n + r]);
        if (!si->floating) memset(fr[n + r].v, 0, sizeof fr[n + r].v);
    }
    for (int i = 0; i < n; i++) {
        const mppi_body_t *b = &m->bodies[i];
        memcpy(fr[i].R, k.Rw[i], sizeof fr[i].R);
        memcpy(fr[i].p, k.pw[i], sizeof fr[i].p);
        real ax[3] = {(real)b->axis[0], (real)b->axis[1], (real)b->axis[2]}, aw[3], t[3], Sw[6];
        m3_vec(fr[i].R, ax, aw);
        if (b->jtype == MPPI_JOINT_REVOLUTE) { cross3(fr[i].p, aw, t); for (int j = 0; j < 3; j++) { Sw[j] = aw[j]; Sw[3 + j] = t[j]; } }
        else for (int j = 0; j < 3; j++) { Sw[j] = 0; Sw[3 + j] = aw[j]; }
        const real *vp = b->parent < 0 ? fr[n + (-1 - b->parent)].v : fr[b->parent].v;
        for (int j = 0; j < 6; j++) fr[i].v[j] = vp[j] + Sw[j] * qd[i];
    }
    for (int f = 0; f < si->n_free; f++) root_frame(root + 13 * si->free_actor[f], &fr[n + si->nb + f]);
}

/* One simulator step of a contact scene; root [A][13] (robot base + free actors) is updated in place.
 * cf (optional) receives the net contact force per rigid body of the last substep. */
void orc_scene_step(const mppi_model_t *m, real *root, real *q, real *qd, const real *target, real *cf_out) {
    scene_info_t si;
    scene_info(m, &si);
    int n = m->n_bodies;
    real h = (real)(m->dt / m->substeps), kd = drive_damping(m, h), kp = m->drive_mode == MPPI_DRIVE_POSITION ? (real)m->drive_kp : 0;
    drive_teleport(m, q, qd, target);
    frame_t *fr = (frame_t *)calloc(NFMAX, sizeof(frame_t));
    real *cf = (real *)calloc(3 * (size_t)m->n_rb + 3, sizeof(real));
    for (int s = 0; s < m->substeps; s++) {
        scene_frames(m, &si, root, q, qd, fr);
        light_pair_t lp[MPPI_MAX_PAIRS];
        int n_lp = 0;
        scene_contacts(m, &si, fr, root, cf, lp, &n_lp);
        real ff[NBMAX], vs[NBMAX], tau[NBMAX], kdh[NBMAX], qdd[NBMAX], abase[NBASEMAX][6];
        for (int i = 0; i < n; i++) {
            ff[i] = m->drive_mode == MPPI_DRIVE_EFFORT ? target[i] : (m->drive_mode == MPPI_DRIVE_POSITION ? kp * (target[i] - q[i]) : 0);
            vs[i] = m->drive_mode == MPPI_DRIVE_VELOCITY ? target[i] : 0;
            tau[i] = ff[i] + kd * (vs[i] - qd[i]);
            kdh[i] = kd * h;
        }
        scene_aba(m, &si, fr, qd, tau, kdh, qdd, abase);
        int any = 0;
        uint32_t satmask = 0;
        for (int i = 0; i < n; i++) {
            real lim = (real)m->bodies[i].effort;
            real tt = ff[i] + kd * (vs[i] - qd[i] - h * qdd[i]);
            if (lim > 0 && (real)fabs((double)tt) > lim) { any = 1; tau[i] = tt > 0 ? lim : -lim; kdh[i] = 0; satmask |= (1u << i) | (tt > 0 ? (1u << (16 + i)) : 0u); }
        }
        if (g_sat_log) g_sat_log[g_sat_n++] = satmask;
        if (any) scene_aba(m, &si, fr, qd, tau, kdh, qdd, abase);
        real qd_start[NBMAX];
        for (int i = 0; i < n; i++) qd_start[i] = qd[i];
        for (int i = 0; i < n; i++) {
            const mppi_body_t *b = &m->bodies[i];
            real vmax = (real)b->velocity;
            qd[i] += h * qdd[i];
            if (vmax > 0) { if (qd[i] > vmax) qd[i] = vmax; if (qd[i] < -vmax) qd[i] = -vmax; }
            const real qold = q[i];
            q[i] += h * qd[i];
            if (b->limited) joint_limit(qold, &q[i], &qd[i], (real)b->lower, (real)b->upper, h);
        }
        if (si.floating)
            for (int r = 0; r < si.nb; r++) root_integrate(root + 13 * base_actor(m, r), abase[r], h);
        /* light bodies held by robot links (light_pair_t): the links' spatial velocities at the END of the substep - the new joint
         * rates (after the velocity and joint limits) on the joint axes of the substep's poses, a floating base from its integrated
         * root row - minus those at its start, through the pair's effective contact, enter the light body's solve */
        if (n_lp > 0) {
            real vn[NFMAX][6];
            for (int r = 0; r < si.nb; r++) {
                frame_t tmp;
                root_frame(root + 13 * base_actor(m, r), &tmp);
                for (int j = 0; j < 6; j++) vn[n + r][j] = si.floating ? tmp.v[j] : 0;
            }
            for (int i = 0; i < n; i++) {
                const mppi_body_t *b = &m->bodies[i];
                real ax[3] = {(real)b->axis[0], (real)b->axis[1], (real)b->axis[2]}, aw[3], t[3], Sw[6];
                m3_vec(fr[i].R, ax, aw);
                if (b->jtype == MPPI_JOINT_REVOLUTE) { cross3(fr[i].p, aw, t); for (int j = 0; j < 3; j++) { Sw[j] = aw[j]; Sw[3 + j] = t[j]; } }
                else for (int j = 0; j < 3; j++) { Sw[j] = 0; Sw[3 + j] = aw[j]; }
                const real *vp = b->parent < 0 ? vn[n + (-1 - b->parent)] : vn[b->parent];
                for (int j = 0; j < 6; j++) vn[i][j] = vp[j] + Sw[j] * qd[i];
            }
            for (int l = 0; l < n_lp; l++) {
                const light_pair_t *r = &lp[l];
                real dv[6], Cd[6];
                for (int j = 0; j < 6; j++) dv[j] = vn[r->ref][j] - fr[r->ref].v[j];
                m6_vec(r->C, dv, Cd);
                for (int k = 0; k < r->n_rec; k++)
                    for (int j = 0; j < 6; j++) Cd[j] += r->g[k][j] * (qd[r->joint[k]] - qd_start[r->joint[k]]);
                for (int j = 0; j < 6; j++) fr[r->light].f[j] += Cd[j];
            }
        }
        for (int f = 0; f < si.n_free; f++) {
            int a = si.free_actor[f];
            const mppi_actor_t *A = &m->actors[a];
            const frame_t *F = &fr[n + si.nb + f];
            double Io6[6] = {0, 0, 0, 0, 0, 0}, h0[3] = {0, 0, 0};
            if (A->type == MPPI_ACTOR_BOX) {
                double x = A->size[0], y = A->size[1], z = A->size[2];
                Io6[0] = A->mass / 12 * (y * y + z * z); Io6[3] = A->mass / 12 * (x * x + z * z); Io6[5] = A->mass / 12 * (x * x + y * y);
            } else Io6[0] = Io6[3] = Io6[5] = 0.4 * A->mass * A->size[0] * A->size[0];
            real I6[36], hw[3], Iv[6], pA6[6], Cv[6], t3[3], g[3] = {0, 0, 0};
            if (A->gravity) for (int j = 0; j < 3; j++) g[j] = (real)m->gravity[j];
            world_inertia(F->R, F->p, (real)A->mass, h0, Io6, I6, hw);
            m6_vec(I6, F->v, Iv);
            crf(F->v, Iv, pA6);
            m6_vec(F->C, F->v, Cv);
            cross3(hw, g, t3);
            real b6[6];
            for (int j = 0; j < 3; j++) { b6[j] = -(pA6[j] + Cv[j] - F->f[j] - t3[j]); b6[3 + j] = -(pA6[3 + j] + Cv[3 + j] - F->f[3 + j] - (real)A->mass * g[j]); }
            for (int j = 0; j < 36; j++) I6[j] += h * F->C[j];
            solve6(I6, b6);
            root_integrate(root + 13 * a, b6, h);
            {   /* free actors: |w| <= MPPI_MAX_ANGULAR_VELOCITY (Isaac Gym's AssetOptions default, include/mppi_hip.h) */
                real *w = root + 13 * a + 10;
                const real w2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], wm = (real)MPPI_MAX_ANGULAR_VELOCITY;
                if (w2 > wm * wm) { const real sc = wm / (real)sqrt((double)w2); w[0] *= sc; w[1] *= sc; w[2] *= sc; }
            }
        }
    }
    if (cf_out) memcpy(cf_out, cf, sizeof(real) * 3 * m->n_rb);
    free(fr); free(cf);
}

/* ------------------------------------------------------------------ stage costs */
static real clamp1(real x) { return x > 1 ? 1 : (x < -1 ? -1 : x); }

real orc_cost(const mppi_model_t *m, const mppi_cost_t *c, const real *root, const real *q, const real *qd, const real *rb, const real *cf) {
    switch (c->kind) {
    case MPPI_COST_POINT_REACH: {
        /* w_nav * || (x, y) - goal ||, x,y = DOF positions 0 and 1 (mppi_planner_wrapper.py:18-21,35) */
        real gx = c->actor[0] >= 0 ? root[13 * c->actor[0]] : (real)c->w[1];
        real gy = c->actor[0] >= 0 ? root[13 * c->actor[0] + 1] : (real)c->w[2];
        real dx = q[0] - gx, dy = q[1] - gy;
        return (real)c->w[0] * (real)sqrt((double)(dx * dx + dy * dy));
    }
    case MPPI_COST_PANDA_REACH: {
        /* examples/panda/planner.py:22-40.  r_pos[:,3:7] is xyzw but pytorch3d reads (r,i,j,k):
         * the reference evaluates ZYX Euler angles of the PERMUTED quaternion; restated as is. */
        const real *ee = rb + 13 * c->link[0];
        const real *g = root + 13 * c->actor[0];
        real dx = ee[0] - g[0], dy = ee[1] - g[1], dz = ee[2] - g[2];
        real dist = (real)sqrt((double)(dx * dx + dy * dy + dz * dz));
        real r = ee[3], i = ee[4], j = ee[5], kk = ee[6];
        real two_s = 2 / (r * r + i * i + j * j + kk * kk);
        real M00 = 1 - two_s * (j * j + kk * kk);
        real M10 = two_s * (i * j + kk * r);
        real M20 = two_s * (i * kk - j * r);
        real a0 = (real)atan2((double)M10, (double)M00);
        real a1 = (real)asin((double)clamp1(-M20));
        real ori = (real)sqrt((double)(a0 * a0 + a1 * a1));
        return (real)c->w[0] * dist + (real)c->w[1] * ori;
    }
    case MPPI_COST_BOXER_PUSH: {
        /* examples/boxer_push/planner.py:26-67.  link[0] = ee_link, actor[0] = block, actor[1] = goal,
         * link[1], link[2] = rigid bodies of paper_obst1 / paper_obst2; w = {robot_to_block, block_to_goal,
         * block_to_goal_ort, push_align, velocity, collision, goal_yaw} */
        const real *r = rb + 13 * c->link[0];
        const real *blk = root + 13 * c->actor[0], *goal = root + 13 * c->actor[1];
        real rbx = r[0] - blk[0], rby = r[1] - blk[1], bgx = goal[0] - blk[0], bgy = goal[1] - blk[1];
        real d_rb = (real)sqrt((double)(rbx * rbx + rby * rby)), d_bg = (real)sqrt((double)(bgx * bgx + bgy * bgy));
        const real *qq = blk + 3; /* quaternion_to_yaw, mppiisaac/utils/conversions.py:4-11 */
        real yaw = (real)atan2((double)(2 * (qq[3] * qq[2] + qq[0] * qq[1])), (double)(qq[3] * qq[3] + qq[0] * qq[0] - qq[1] * qq[1] - qq[2] * qq[2]));
        real ort = (real)fabs((double)(yaw - (real)c->w[6]));
        real align = (rbx * bgx + rby * bgy) / (d_rb * d_bg) + 1;
        real coll = 0;
        if (cf) coll = (real)(fabs((double)cf[3 * c->link[1]]) + fabs((double)cf[3 * c->link[1] + 1]) + fabs((double)cf[3 * c->link[2]]) + fabs((double)cf[3 * c->link[2] + 1]));
        real vel = (real)sqrt((double)(blk[7] * blk[7] + blk[8] * blk[8]));
        return (real)c->w[0] * d_rb + (real)c->w[1] * d_bg + (real)c->w[2] * ort + (real)c->w[3] * align + (real)c->w[4] * vel + (real)c->w[5] * coll;
    }
    case MPPI_COST_PANDA_PICK: {
        /* examples/panda_pick/planner.py:24-53.  link[0] = panda_ee, link[1] = table body, actor[0] = block,
         * actor[1] = goal; w = {robot_to_block, block_to_goal, collision, robot_ori} */
        const real *ee = rb + 13 * c->link[0];
        const real *blk = root + 13 * c->actor[0], *goal = root + 13 * c->actor[1];
        real d1 = 0, d2 = 0;
        for (int j = 0; j < 3; j++) { d1 += (ee[j] - blk[j]) * (ee[j] - blk[j]); d2 += (blk[j] - goal[j]) * (blk[j] - goal[j]); }
        real forces = 0;
        if (cf) for (int j = 0; j < 3; j++) forces += (real)fabs((double)cf[3 * c->link[1] + j]);
        real r = ee[3], i = ee[4], j = ee[5], kk = ee[6];
        real two_s = 2 / (r * r + i * i + j * j + kk * kk);
        real M00 = 1 - two_s * (j * j + kk * kk), M10 = two_s * (i * j + kk * r), M20 = two_s * (i * kk - j * r);
        real a0 = (real)atan2((double)M10, (double)M00), a1 = (real)asin((double)clamp1(-M20));
        return (real)c->w[0] * (real)sqrt((double)d1) + (real)c->w[1] * (real)sqrt((double)d2) + (real)c->w[2] * forces
               + (real)c->w[3] * (real)sqrt((double)(a0 * a0 + a1 * a1));
    }
    case MPPI_COST_PROGRAM: {
        /* a weighted sum of the measurements every Objective of the reference's examples/<x>/planner.py is built from
         * (include/mppi_hip.h MPPI_OP_*): distances between link / actor positions, the tilt of a link (the ZYX-Euler quirk
         * above), yaw error, push alignment, contact-force L1 norms, actor speed, DOF terms, height terms.  rb rows cover
         * every rigid body of the env (robot links and box / sphere bodies), root rows every actor. */
        real total = 0;
        int n_dof = m->n_bodies;
        for (int it = 0; it < c->n_terms; it++) {
            const mppi_term_t *t = &c->terms[it];
            real P[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            for (int a = 0; a < 3; a++) {
                const real *src = NULL;
                if (t->src[a] == MPPI_SRC_RB) src = rb + 13 * t->idx[a];
                else if (t->src[a] == MPPI_SRC_ACTOR) src = root + 13 * t->idx[a];
                if (src) for (int j = 0; j < 3; j++) P[a][j] = src[j];
                else if (t->src[a] == MPPI_SRC_DOF_XY) { P[a][0] = q[0]; P[a][1] = q[1]; }
                else if (t->src[a] == MPPI_SRC_CONST) for (int j = 0; j < 3; j++) P[a][j] = (real)t->p[j];
            }
            real v = 0;
            switch (t->op) {
            case MPPI_OP_DIST: {
                double s2 = 0;
                for (int j = 0; j < t->n; j++) s2 += (double)(P[0][j] - P[1][j]) * (double)(P[0][j] - P[1][j]);
                v = (real)sqrt(s2);
                break;
            }
            case MPPI_OP_TILT: {
                const real *ee = rb + 13 * t->idx[0];
                real r = ee[3], i = ee[4], j = ee[5], kk = ee[6];
                real two_s = 2 / (r * r + i * i + j * j + kk * kk);
                real M00 = 1 - two_s * (j * j + kk * kk), M10 = two_s * (i * j + kk * r), M20 = two_s * (i * kk - j * r);
                real a0 = (real)atan2((double)M10, (double)M00), a1 = (real)asin((double)clamp1(-M20));
                v = (real)sqrt((double)(a0 * a0 + a1 * a1));
                break;
            }
            case MPPI_OP_YAW_ABS: {
                const real *qq = root + 13 * t->idx[0] + 3;
                real yaw = (real)atan2((double)(2 * (qq[3] * qq[2] + qq[0] * qq[1])), (double)(qq[3] * qq[3] + qq[0] * qq[0] - qq[1] * qq[1] - qq[2] * qq[2]));
                v = (real)fabs((double)(yaw - (real)t->p[3]));
                break;
            }
            case MPPI_OP_ALIGN: {
                real ax = P[0][0] - P[1][0], ay = P[0][1] - P[1][1], cx = P[2][0] - P[1][0], cy = P[2][1] - P[1][1];
                v = (ax * cx + ay * cy) / ((real)sqrt((double)(ax * ax + ay * ay)) * (real)sqrt((double)(cx * cx + cy * cy))) + 1;
                break;
            }
            case MPPI_OP_FORCE_L1:
                if (cf) for (int j = 0; j < t->n; j++) v += (real)fabs((double)cf[3 * t->idx[0] + j]);
                break;
            case MPPI_OP_SPEED: {
                const real *vv = root + 13 * t->idx[0] + 7;
                double s2 = 0;
                for (int j = 0; j < t->n; j++) s2 += (double)vv[j] * (double)vv[j];
                v = (real)sqrt(s2);
                break;
            }
            case MPPI_OP_DOF_SQ:
                for (int i = t->idx[0]; i < t->idx[1] && i < n_dof; i++) {
                    real x = (t->n == 0 ? q[i] : qd[i]) - (i - t->idx[0] < t->idx[2] ? (real)(float)t->p[i - t->idx[0]] : 0); /* fp32 constants, as the planners hold them */
                    v += x * x;
                }
                break;
            case MPPI_OP_ABS_DZ: v = (real)fabs((double)(P[0][2] - P[1][2])); break;
            case MPPI_OP_BELOW: v = (real)t->p[3] - P[0][2]; if (v < 0) v = 0; break;
            default: break;
            }
            total += (real)t->w * v;
        }
        return total;
    }
    default:
        return 0;
    }
}

/* ------------------------------------------------------------------ halton-spline sampler */
static const int PRIMES[MPPI_MAX_KNOTS * MPPI_MAX_NU] = {
    2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113,
    127, 131, 137, 139, 149, 151, 157, 163, 167, 173, 179, 181, 191, 193, 197, 199, 211, 223, 227, 229, 233, 239, 241, 251,
    257, 263, 269, 271, 277, 281, 283, 293, 307, 311, 313, 317, 331, 337, 347, 349, 353, 359, 367, 373, 379, 383, 389, 397,
    401, 409, 419, 421, 431, 433, 439, 443, 449, 457, 461, 463, 467, 479, 487, 491, 499, 503, 509, 521, 523, 541, 547, 557,
    563, 569, 571, 577, 587, 593, 599, 601, 607, 613, 617, 619, 631, 641, 643, 647, 653, 659, 661, 673, 677, 683, 691, 701,
    709, 719, 727, 733, 739, 743, 751, 757, 761, 769, 773, 787, 797, 809, 811, 821, 823, 827, 829, 839, 853, 857, 859, 863,
    877, 881, 883, 887, 907, 911, 919, 929, 937, 941, 947, 953, 967, 971, 977, 983, 991, 997, 1009, 1013, 1019, 1021, 1031,
    1033, 1039, 1049, 1051, 1061, 1063, 1069, 1087, 1091, 1093, 1097, 1103, 1109, 1117, 1123, 1129, 1151, 1153, 1163};

/* radical inverse of n in base p with the linear digit scramble digit -> (digit * mult) mod p
 * (mult = 1: the plain van der Corput / Halton sequence, KAT'ed against scipy.stats.qmc.Halton(scramble=False)) */
double orc_radical_inverse(uint32_t n, uint32_t p, uint32_t mult) {
    double f = 1.0 / p, r = 0.0;
    while (n > 0) {
        uint32_t dgt = n % p;
        r += f * (double)((dgt * mult) % p);
        n /= p;
        f /= p;
    }
    return r;
}
int orc_prime(int dim) { return PRIMES[dim]; }
/* the sampler's sequence: mult = round(0.618 p) */
double orc_halton(uint32_t n, int dim) {
    int p = PRIMES[dim];
    int mult = (int)(0.6180339887498949 * p + 0.5);
    if (mult < 1) mult = 1;
    return orc_radical_inverse(n, (uint32_t)p, (uint32_t)mult);
}

/* inverse standard normal CDF: Acklam's rational approximation + one Halley step (double accuracy) */
double orc_norminv(double p) {
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    double x, q, r;
    if (p > 0.5) return -orc_norminv(1 - p); /* 1 - p is exact for p >= 0.5: the refinement below keeps its accuracy in the upper tail */
    if (p < 0.02425) {
        q = sqrt(-2 * log(p));
        x = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    } else if (p <= 1 - 0.02425) {
        q = p - 0.5; r = q * q;
        x = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q / (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1);
    } else {
        q = sqrt(-2 * log(1 - p));
        x = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    }
    for (int it = 0; it < 2; it++) {
        double e = 0.5 * erfc(-x / sqrt(2.0)) - p;
        double u = e * sqrt(2 * M_PI) * exp(x * x / 2);
        x = x - u / (1 + x * u / 2);
    }
    return x;
}

/* eps[t][c][k] = sigma_c * sum_i B[t][i] * Phi^-1( halton(g + 1 + index_base, i*nu + c) ),  g = k_offset + k */
void orc_sample(const mppi_config_t *cfg, uint32_t index_base, real *eps) {
    int K = cfg->num_samples, H = cfg->horizon, nu = cfg->nu, nk = cfg->n_knots;
    double *z = (double *)malloc(sizeof(double) * nk * nu);
    for (int k = 0; k < K; k++) {
        uint32_t n = (uint32_t)(cfg->k_offset + k) + 1u + index_base;
        for (int dmn = 0; dmn < nk * nu; dmn++) z[dmn] = orc_norminv(orc_halton(n, dmn));
        for (int t = 0; t < H; t++)
            for (int c = 0; c < nu; c++) {
                double s = 0;
                for (int i = 0; i < nk; i++) s += cfg->spline_basis[t * nk + i] * z[i * nu + c];
                eps[((size_t)t * nu + c) * K + k] = (real)(sqrt(cfg->noise_sigma_diag[c]) * s);
            }
    }
    free(z);
}

/* ------------------------------------------------------------------ counter-based Gaussian sampler
 * MPPI_SAMPLE_NORMAL (include/mppi_hip.h): what mppi_torch's "simple" mode / the random knot source of its
 * halton-spline mode draw from torch's generator inside MPPIPlanner.command (call sites reference
 * mppiisaac/planner/mppi_isaac.py:84,113; configs reference conf/mppi/omnipanda_effort.yaml:4-5) is drawn here from
 * Philox4x32-10 (Salmon et al. SC'11; known-answer vectors of the Random123 distribution in tests/test_oracle_kat.py)
 * so that every shard, every GPU and this oracle see the same noise for a (seed, iteration). */
void orc_philox4x32(const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
/* standard normal knot i of control c of global sample g: words (2p, 2p+1) of the block i/4 -> Box-Muller */
double orc_normal_knot(uint32_t seed, uint32_t iteration, uint32_t g, int c, int i) {
    uint32_t ctr[4] = {g, (uint32_t)c, (uint32_t)(i / 4), iteration}, key[2] = {seed, 0x4D505049u}, x[4];
    orc_philox4x32(ctr, key, x);
    int p = (i % 4) / 2, e = i % 2;
    double ua = ((double)x[2 * p] + 0.5) / 4294967296.0, ub = ((double)x[2 * p + 1] + 0.5) / 4294967296.0;
    double r = sqrt(-2.0 * log(ua)), a = 6.283185307179586 * ub;
    return e == 0 ? r * cos(a) : r * sin(a);
}
/* eps[t][c][k] = mu_c + sigma_c * sum_i B[t][i] z_i  (n_knots == H: eps_t = mu_c + sigma_c z_t) */
void orc_sample_normal(const mppi_config_t *cfg, uint32_t iteration, real *eps) {
    int K = cfg->num_samples, H = cfg->horizon, nu = cfg->nu, nk = cfg->n_knots;
    double *z = (double *)malloc(sizeof(double) * nk);
    for (int k = 0; k < K; k++)
        for (int c = 0; c < nu; c++) {
            uint32_t g = (uint32_t)(cfg->k_offset + k);
            double sg = sqrt(cfg->noise_sigma_diag[c]), mu = cfg->noise_mu[c];
            for (int i = 0; i < nk; i++) z[i] = orc_normal_knot((uint32_t)cfg->seed, iteration, g, c, i);
            for (int t = 0; t < H; t++) {
                double s = 0;
                if (nk == H) s = z[t];
                else for (int i = 0; i < nk; i++) s += cfg->spline_basis[t * nk + i] * z[i];
                eps[((size_t)t * nu + c) * K + k] = (real)(mu + sg * s);
            }
        }
    free(z);
}

/* ------------------------------------------------------------------ per-sample actor randomisation */
/* The reference draws, per env and per noisy actor, a size (normal), a mass and a friction (uniform +-percentage)
 * from numpy's unseeded global generator (isaacgym_wrapper.py:430-475, isaacgym_utils.py:30-52). Here the draws
 * are a counter-based hash of (seed, global sample index, actor, component) so that every GPU shard, the host
 * emulation and this oracle see the same per-sample world. fp32 arithmetic on purpose: the draws are inputs. */
static float hash_uniform(int seed, int g, int actor, int k) {
    uint32_t h = (uint32_t)seed * 0x9E3779B1u ^ (uint32_t)(g + 1) * 0x85EBCA77u ^ (uint32_t)(actor + 1) * 0xC2B2AE3Du ^ (uint32_t)(k + 1) * 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
/* [actor][5]: size deltas xyz, mass scale, friction of sample g */
void orc_randomise_draws(const mppi_model_t *m, int g, double *out) {
    for (int a = 0; a < m->n_actors; a++) {
        const mppi_actor_t *A = &m->actors[a];
        int on = m->randomize_seed >= 0 && a != m->robot_actor;
        for (int j = 0; j < 3; j++) {
            double sg = on ? A->noise_sigma_size[j] : 0;
            out[5 * a + j] = sg != 0 ? sg * orc_norminv((double)hash_uniform(m->randomize_seed, g, a, j)) : 0;
        }
        double pm = on ? A->noise_percentage_mass : 0, pf = on ? A->noise_percentage_friction : 0;
        out[5 * a + 3] = pm != 0 ? 1 + pm * (2 * (double)hash_uniform(m->randomize_seed, g, a, 3) - 1) : 1;
        out[5 * a + 4] = pf != 0 ? A->friction * (1 + pf * (2 * (double)hash_uniform(m->randomize_seed, g, a, 4) - 1)) : A->friction;
    }
}
/* the model sample g simulates: noisy actors and their shapes take this sample's size / mass / friction */
void orc_randomise_model(const mppi_model_t *m, int g, mppi_model_t *out) {
    *out = *m;
    if (m->randomize_seed < 0) return;
    double dr[5 * MPPI_MAX_ACTORS];
    orc_randomise_draws(m, g, dr);
    for (int a = 0; a < m->n_actors; a++) {
        if (a == m->robot_actor) continue;
        mppi_actor_t *A = &out->actors[a];
        if (A->type == MPPI_ACTOR_BOX) for (int j = 0; j < 3; j++) A->size[j] += dr[5 * a + j];
        else A->size[0] += dr[5 * a];
        A->mass *= dr[5 * a + 3];
        A->friction = dr[5 * a + 4];
    }
    for (int i = 0; i < m->n_shapes; i++) {
        mppi_shape_t *S = &out->shapes[i];
        if (S->actor == m->robot_actor) continue;
        if (S->type == MPPI_SHAPE_BOX) for (int j = 0; j < 3; j++) S->size[j] += 0.5 * dr[5 * S->actor + j];
        else if (S->type == MPPI_SHAPE_SPHERE) S->size[0] += dr[5 * S->actor];
        S->friction = dr[5 * S->actor + 4];
    }
    out->randomize_seed = -1;
}

/* ------------------------------------------------------------------ rollout + update */
/* One sample: returns total cost S: sum_t gamma^t c_t + lambda * sum_t U_t^T Sigma^-1 du_t */
static real rollout_one(const mppi_model_t *m_nominal, const mppi_config_t *cfg, const mppi_cost_t *cost, const real *dof0,
                        const real *root0, const real *U, const real *eps, const real *prior, int k, real *du, real *viz) {
    int K = cfg->num_samples, H = cfg->horizon, nu = cfg->nu, n = m_nominal->n_bodies;
    int g = cfg->k_offset + k;
    const mppi_model_t *m = m_nominal;
    mppi_model_t *mine = NULL;
    if (m_nominal->randomize_seed >= 0) {
        mine = (mppi_model_t *)malloc(sizeof(mppi_model_t));
        orc_randomise_model(m_nominal, g, mine);
        m = mine;
    }
    real q[NBMAX], qd[NBMAX], target[NBMAX], u[MPPI_MAX_NU];
    real *rb = (real *)malloc(sizeof(real) * 13 * m->n_rb);
    real *cf = (real *)calloc(3 * (size_t)m->n_rb + 3, sizeof(real));
    real root[13 * MPPI_MAX_ACTORS]; /* per-sample root rows (robot base and free actors move in contact scenes) */
    memcpy(root, root0, sizeof(real) * 13 * m->n_actors);
    const int scene = orc_is_scene(m);
    for (int i = 0; i < n; i++) { q[i] = dof0[2 * i]; qd[i] = dof0[2 * i + 1]; }
    real S = 0, ctrl = 0, disc = 1;
    for (int t = 0; t < H; t++) {
        for (int c = 0; c < nu; c++) {
            real v = U[t * nu + c] + eps[((size_t)t * nu + c) * K + k];
            if (cfg->sample_null_action && g == cfg->k_total - 1) v = 0;
            if (cfg->use_priors && prior && g == cfg->k_total - 2) v = prior[t * nu + c];
            real lo = (real)cfg->u_min[c], hi = (real)cfg->u_max[c];
            v = v < lo ? lo : (v > hi ? hi : v);
            u[c] = v;
            real d = v - U[t * nu + c];
            du[((size_t)t * nu + c) * K + k] = d;
            real term = U[t * nu + c] * d / (real)cfg->noise_sigma_diag[c];
            ctrl += (real)cfg->lambda_ * (cfg->noise_abs_cost ? (real)fabs((double)term) : term);
        }
        orc_cmd_map(m, u, target);
        if (scene) orc_scene_step(m, root, q, qd, target, cf);
        else orc_step(m, root, q, qd, target);
        orc_rigid_body_state(m, root, q, qd, rb, NULL);
        real ct = orc_cost(m, cost, root, q, qd, rb, cf);
        S += disc * ct;
        disc *= (real)cfg->rollout_var_discount;
        if (viz && cfg->want_rollouts) {
            const real *o = rb + 13 * (m->actors[m->robot_actor].first_rb + cfg->viz_link);
            for (int j = 0; j < 3; j++) viz[((size_t)t * K + k) * 3 + j] = o[j];
        }
    }
    free(rb); free(cf); free(mine);
    return S + ctrl;
}

void orc_rollout(const mppi_model_t *m, const mppi_config_t *cfg, const mppi_cost_t *cost, const real *dof0, const real *root0,
                 const real *U, const real *eps, const real *prior, real *S, real *du, real *viz) {
    int K = cfg->num_samples;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; k++) S[k] = rollout_one(m, cfg, cost, dof0, root0, U, eps, prior, k, du, viz);
}

/* shard record: beta = min S, eta = sum exp(-(S-beta)/lambda), N[t][c] = sum w du    */
void orc_record(const mppi_config_t *cfg, const real *S, const real *du, real *rec) {
    int K = cfg->num_samples, HN = cfg->horizon * cfg->nu;
    real beta = INFINITY;
    for (int k = 0; k < K; k++) if (isfinite((double)S[k]) && S[k] < beta) beta = S[k];
    real eta = 0;
    for (int j = 0; j < HN; j++) rec[2 + j] = 0;
    for (int k = 0; k < K; k++) {
        if (!isfinite((double)S[k])) continue; /* NaN/Inf cost -> weight 0 */
        real w = (real)exp(-(double)(S[k] - beta) / cfg->lambda_);
        eta += w;
        for (int j = 0; j < HN; j++) rec[2 + j] += w * du[(size_t)j * K + k];
    }
    rec[0] = beta; rec[1] = eta;
}

/* combine records, U += N/eta, action = U[0], shift left, append u_init */
void orc_update(const mppi_config_t *cfg, const real *recs, int nrec, real *U, real *action, real *beta_eta) {
    int H = cfg->horizon, nu = cfg->nu, HN = H * nu, RF = 2 + HN;
    real beta = INFINITY, eta = 0;
    for (int r = 0; r < nrec; r++) if (recs[r * RF] < beta) beta = recs[r * RF];
    real *N = (real *)calloc(HN, sizeof(real));
    for (int r = 0; r < nrec; r++) {
        if (!(recs[r * RF + 1] > 0)) continue;
        real sc = (real)exp(-(double)(recs[r * RF] - beta) / cfg->lambda_);
        eta += sc * recs[r * RF + 1];
        for (int j = 0; j < HN; j++) N[j] += sc * recs[r * RF + 2 + j];
    }
    for (int j = 0; j < HN; j++) U[j] += N[j] / eta;
    for (int c = 0; c < nu; c++) action[c] = U[c];
    for (int t = 0; t + 1 < H; t++) for (int c = 0; c < nu; c++) U[t * nu + c] = U[(t + 1) * nu + c];
    for (int c = 0; c < nu; c++) U[(H - 1) * nu + c] = (real)cfg->u_init;
    if (beta_eta) { beta_eta[0] = beta; beta_eta[1] = eta; }
    free(N);
}

/* full control iteration for one shard: the timed "reference-structured CPU pipeline" */
void orc_command(const mppi_model_t *m, const mppi_config_t *cfg, const mppi_cost_t *cost, const real *dof0, const real *root0,
                 real *U, const real *eps, real *S, real *du, real *action) {
    int RF = 2 + cfg->horizon * cfg->nu;
    real *rec = (real *)malloc(sizeof(real) * RF);
    orc_rollout(m, cfg, cost, dof0, root0, U, eps, NULL, S, du, NULL);
    orc_record(cfg, S, du, rec);
    orc_update(cfg, rec, 1, U, action, NULL);
    free(rec);
}

/* ------------------------------------------------------------------ batched env step (CPU pipeline baseline)
 * What `apply_robot_cmd` + `step` + the four tensor refreshes are to the reference's Python horizon loop
 * (mppiisaac/planner/mppi_isaac.py:57-65, isaacgym_wrapper.py:524-572,639-655): K independent envs advanced by one dt
 * with per-env commands u [K][nu], their reference-layout state rows rewritten in place: dof [K][2n] interleaved,
 * root [K][A][13], rigid bodies [K][B][13], net contact forces [K][B][3].  OpenMP over the envs, as Isaac Gym's CPU
 * pipeline threads over them; orc_set_threads picks the thread count (bench.py times 1 and all host cores). */
#include <omp.h>
void orc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int orc_get_max_threads(void) { return omp_get_max_threads(); }
void orc_envs_step(const mppi_model_t *m_nominal, int K, int g0, const real *u, real *dof, real *root, real *rb, real *cf) {
    const int n = m_nominal->n_bodies, A = m_nominal->n_actors, B = m_nominal->n_rb, nu = m_nominal->nu;
    const int scene = orc_is_scene(m_nominal);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; k++) {
        const mppi_model_t *m = m_nominal;
        mppi_model_t *mine = NULL;
        if (m_nominal->randomize_seed >= 0) {
            mine = (mppi_model_t *)malloc(sizeof(mppi_model_t));
            orc_randomise_model(m_nominal, g0 + k, mine);
            m = mine;
        }
        real q[NBMAX], qd[NBMAX], target[NBMAX];
        real *d = dof + (size_t)k * 2 * n, *r = root + (size_t)k * 13 * A;
        for (int i = 0; i < n; i++) { q[i] = d[2 * i]; qd[i] = d[2 * i + 1]; }
        orc_cmd_map(m, u + (size_t)k * nu, target);
        if (scene) orc_scene_step(m, r, q, qd, target, cf + (size_t)k * 3 * B);
        else orc_step(m, r, q, qd, target);
        for (int i = 0; i < n; i++) { d[2 * i] = q[i]; d[2 * i + 1] = qd[i]; }
        orc_rigid_body_state(m, r, q, qd, rb + (size_t)k * 13 * B, NULL);
        free(mine);
    }
}

/* diagnostics: the rollout of ONE sample k with its per-substep saturation sets written to log[H * substeps] */
real orc_rollout_satlog(const mppi_model_t *m, const mppi_config_t *cfg, const mppi_cost_t *cost, const real *dof0, const real *root0,
                        const real *U, const real *eps, int k, uint32_t *log) {
    real *du = (real *)calloc((size_t)cfg->horizon * cfg->nu * cfg->num_samples, sizeof(real));
    g_sat_log = log; g_sat_n = 0;
    real S = rollout_one(m, cfg, cost, dof0, root0, U, eps, NULL, k, du, NULL);
    g_sat_log = NULL;
    free(du);
    return S;
}

int orc_sizeof_real(void) { return (int)sizeof(real); }
int orc_sizeof_model(void) { return (int)sizeof(mppi_model_t); }
int orc_sizeof_config(void) { return (int)sizeof(mppi_config_t); }
int orc_sizeof_cost(void) { return (int)sizeof(mppi_cost_t); }
