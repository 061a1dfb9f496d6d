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
 * state packing, quaternion_to_yaw) is pinned by tests/golden/ (tools/make_golden.py).
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
    real Rb[9], pb[3];               /* base->world                                      */
    real X[NBMAX][36];               /* Plucker parent->child                            */
    real S[NBMAX][6];
    real v[NBMAX][6];                /* spatial velocity, body coordinates               */
    real c[NBMAX][6];
} kin_t;

static void kinematics(const mppi_model_t *m, const real *root, const real *q, const real *qd, kin_t *k) {
    const real *rs = root + 13 * m->robot_actor;
    k->pb[0] = rs[0]; k->pb[1] = rs[1]; k->pb[2] = rs[2];
    quat_to_R(rs + 3, k->Rb);
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
        const real *Rp = b->parent < 0 ? k->Rb : k->Rw[b->parent];
        const real *pp = b->parent < 0 ? k->pb : k->pw[b->parent];
        real t[3];
        m3_mul(Rp, k->Rj[i], k->Rw[i]);
        m3_vec(Rp, k->pj[i], t);
        for (int j = 0; j < 3; j++) k->pw[i][j] = pp[j] + t[j];
        /* v_i = X v_parent + S qd ; c_i = v_i x (S qd)    (base is fixed: v_base = 0) */
        real vj[6];
        for (int j = 0; j < 6; j++) vj[j] = k->S[i][j] * qd[i];
        if (b->parent < 0) memcpy(k->v[i], vj, sizeof vj);
        else { m6_vec(k->X[i], k->v[b->parent], k->v[i]); for (int j = 0; j < 6; j++) k->v[i][j] += vj[j]; }
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
        m3_tvec(k->Rb, g, gb);
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

/* One simulator step of dt = substeps * h (IsaacGymWrapper.step, isaacgym_wrapper.py:639-645). */
void orc_step(const mppi_model_t *m, const real *root, real *q, real *qd, const real *target) {
    int n = m->n_bodies;
    real h = (real)(m->dt / m->substeps), kd = (real)m->drive_kd;
    for (int s = 0; s < m->substeps; s++) {
        kin_t k;
        kinematics(m, root, q, qd, &k);
        real ff[NBMAX], vs[NBMAX], tau[NBMAX], kdh[NBMAX], qdd[NBMAX];
        for (int i = 0; i < n; i++) {
            ff[i] = m->drive_mode == MPPI_DRIVE_EFFORT ? target[i] : 0;
            vs[i] = m->drive_mode == MPPI_DRIVE_VELOCITY ? target[i] : 0;

            tau[i] = ff[i] + kd * (vs[i] - qd[i]);
            kdh[i] = kd * h;
        }
        aba_solve(m, &k, tau, kdh, qdd);
        /* drive-force clamp (URDF <limit effort>): joints whose implicit drive force exceeds the
         * limit are re-solved with the constant saturated force (one re-solve) */
        int any = 0;
        for (int i = 0; i < n; i++) {
            real lim = (real)m->bodies[i].effort;
            real tt = ff[i] + kd * (vs[i] - qd[i] - h * qdd[i]);
            if (lim > 0 && (real)fabs((double)tt) > lim) { any = 1; tau[i] = tt > 0 ? lim : -lim; kdh[i] = 0; }
        }
        if (any) aba_solve(m, &k, tau, kdh, qdd);
        for (int i = 0; i < n; i++) {
            const mppi_body_t *b = &m->bodies[i];
            real vmax = (real)b->velocity;
            qd[i] += h * qdd[i];
            if (vmax > 0) { if (qd[i] > vmax) qd[i] = vmax; if (qd[i] < -vmax) qd[i] = -vmax; }
            q[i] += h * qd[i];
            if (b->limited) {
                if (q[i] < (real)b->lower) { q[i] = (real)b->lower; if (qd[i] < 0) qd[i] = 0; }
                if (q[i] > (real)b->upper) { q[i] = (real)b->upper; if (qd[i] > 0) qd[i] = 0; }
            }
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
            memcpy(rb + 13 * A->first_rb, root + 13 * a, 13 * sizeof(real));
            continue;
        }
        for (int l = 0; l < m->n_links; l++) {
            const mppi_link_t *L = &m->links[l];
            real Rl[9], pl[3], Rw[9], pw[3], t[3], quat[4], wv[3] = {0, 0, 0}, lv[3] = {0, 0, 0};
            for (int j = 0; j < 9; j++) Rl[j] = (real)L->R[j];
            for (int j = 0; j < 3; j++) pl[j] = (real)L->p[j];
            const real *Rbw = L->body < 0 ? k.Rb : k.Rw[L->body];
            const real *pbw = L->body < 0 ? k.pb : k.pw[L->body];
            m3_mul(Rbw, Rl, Rw);
            m3_vec(Rbw, pl, t);
            for (int j = 0; j < 3; j++) pw[j] = pbw[j] + t[j];
            if (L->body >= 0) { /* velocity of the link origin: R_w (v + w x p_l) */
                const real *v = k.v[L->body];
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

/* ------------------------------------------------------------------ stage costs */
static real clamp1(real x) { return x > 1 ? 1 : (x < -1 ? -1 : x); }

real orc_cost(const mppi_model_t *m, const mppi_cost_t *c, const real *root, const real *q, const real *qd, const real *rb) {
    (void)m; (void)qd;
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

/* linearly digit-scrambled radical inverse: digit -> (digit * mult) mod p, mult = round(0.618 p) */
double orc_halton(uint32_t n, int dim) {
    int p = PRIMES[dim];
    int mult = (int)(0.6180339887498949 * p + 0.5);
    if (mult < 1) mult = 1;
    double f = 1.0 / p, r = 0.0;
    while (n > 0) {
        uint32_t dgt = n % (uint32_t)p;
        r += f * (double)((dgt * (uint32_t)mult) % (uint32_t)p);
        n /= (uint32_t)p;
        f /= p;
    }
    return r;
}

/* inverse standard normal CDF: Acklam's rational approximation + one Halley step (double accuracy) */
double orc_norminv(double p) {
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01};
    static const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    double x, q, r;
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

/* ------------------------------------------------------------------ rollout + update */
/* One sample: returns total cost S: sum_t gamma^t c_t + lambda * sum_t U_t^T Sigma^-1 du_t */
static real rollout_one(const mppi_model_t *m, const mppi_config_t *cfg, const mppi_cost_t *cost, const real *dof0,
                        const real *root0, const real *U, const real *eps, const real *prior, int k, real *du, real *viz) {
    int K = cfg->num_samples, H = cfg->horizon, nu = cfg->nu, n = m->n_bodies;
    int g = cfg->k_offset + k;
    real q[NBMAX], qd[NBMAX], target[NBMAX], u[MPPI_MAX_NU];
    real *rb = (real *)malloc(sizeof(real) * 13 * m->n_rb);
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
        orc_step(m, root0, q, qd, target);
        orc_rigid_body_state(m, root0, q, qd, rb, NULL);
        real ct = orc_cost(m, cost, root0, q, qd, rb);
        S += disc * ct;
        disc *= (real)cfg->rollout_var_discount;
        if (viz && cfg->want_rollouts) {
            const real *o = rb + 13 * (m->actors[m->robot_actor].first_rb + cfg->viz_link);
            for (int j = 0; j < 3; j++) viz[((size_t)t * K + k) * 3 + j] = o[j];
        }
    }
    free(rb);
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

int orc_sizeof_real(void) { return (int)sizeof(real); }
int orc_sizeof_model(void) { return (int)sizeof(mppi_model_t); }
int orc_sizeof_config(void) { return (int)sizeof(mppi_config_t); }
int orc_sizeof_cost(void) { return (int)sizeof(mppi_cost_t); }
