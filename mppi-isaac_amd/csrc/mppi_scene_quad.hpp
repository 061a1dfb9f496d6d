// mppi_scene_quad.hpp - contact-scene step with the articulated-body algebra split over the quad.
//
// k_rollout_scene_quad gives every sample a 4-lane quad.  The contact work was already dealt over the lanes
// (mppi_scene.hpp, kSplitQuad); the robot's kinematics and articulated-body solve were still computed four times,
// once per lane.  Here they use the quad layout of mppi_quad.hpp instead - lane r owns component (matrix row) r of
// every 3-vector / 3x3 block, DPP quad_perm for the cross-lane traffic - which cuts their per-lane instruction
// stream ~3x and their register footprint with it (the one-lane version spills: 256 VGPR + 256 AGPR + scratch).
//
// What changes against quad_aba (fixed base, no contact): per-frame external wrenches and implicit contact
// dampings from the sample's LDS accumulators (lane r gathers ITS rows), explicit gravity, and a floating base
// (the base's 6x6 system is gathered to all lanes and solved by the replicated Cholesky of mppi_scene.hpp).
// Sample state (q, qd, base and free-body rows) stays replicated across the quad, as do the free-body solves.
#pragma once
#include "mppi_quad.hpp"
#include "mppi_scene.hpp"

namespace mppi {

// ---- quad <-> per-sample LDS rows -----------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
template <int R>
__device__ __forceinline__ float qget(QF x) { return bc<R>(x); }
// lane r reads / writes L[base + o_r]
// (Measured and NOT kept, round 3: the optimiser folds the compile-time `base` into the select - every access site becomes an
// address register of its own, ~200 in the gripper scene's kernel, hoisted out of the rollout loops into AGPRs.  Making the
// lane's pick opaque (`asm("" : "+v"(v))`) turns them into ten lane parts plus DS immediate offsets: 229 -> 166 AGPRs, but
// the kernel ran 0.7 % SLOWER at equal state (1.9446 -> 1.9582 ms): the AGPR copies were not what it waits for.)
__device__ __forceinline__ QF qgather(const LMem &L, int base, int o0, int o1, int o2) {
    const int r = quad_row();
    return L[base + (r == 0 ? o0 : (r == 1 ? o1 : o2))];
}
__device__ __forceinline__ void qscatter(const LMem &L, int base, int o0, int o1, int o2, QF x) {
    const int r = quad_row();
    L[base + (r == 0 ? o0 : (r == 1 ? o1 : o2))] = x;
}
#else
template <int R>
inline float qget(QF x) { return x.v[R]; }
inline QF qgather(const LMem &L, int base, int o0, int o1, int o2) { return QF{{L[base + o0], L[base + o1], L[base + o2], L[base + o0]}}; }
inline void qscatter(const LMem &L, int base, int o0, int o1, int o2, QF x) {
    L[base + o0] = x.v[0];
    L[base + o1] = x.v[1];
    L[base + o2] = x.v[2];
}
#endif

// frame row block of mppi_scene.hpp: R[9] row-major, p[3], w[3], vO[3]
MPPI_HD void qframe_store(const LMem &L, int ent, const QM3 &R, QF p, const QSV &v) {
    const int o = ent * 18;
    for (int c = 0; c < 3; c++) qscatter(L, o + c, 0, 3, 6, R.c[c]);  // lane r: R[r][c] -> o + 3r + c
    qscatter(L, o + 9, 0, 1, 2, p);
    qscatter(L, o + 12, 0, 1, 2, v.a);
    qscatter(L, o + 15, 0, 1, 2, v.l);
}

constexpr int sym_index(int r, int c) {  // position of (r, c) in (xx, xy, xz, yy, yz, zz)
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    return lo * 3 - lo * (lo - 1) / 2 + (hi - lo);
}
// external wrench and implicit damping of frame `ent` (accumulator rows of contact_forces) in the quad layout
MPPI_HD void qacc_load(const LMem &L, int base, int ent, QSV &fe, QAI &C) {
    const int o = base + ent * 27;
    fe.a = qgather(L, o, 0, 1, 2);
    fe.l = qgather(L, o + 3, 0, 1, 2);
    for (int j = 0; j < 3; j++) {  // rotated rows: X[j] of lane r = X[r][(r + j) % 3]
        const int c0 = j % 3, c1 = (1 + j) % 3, c2 = (2 + j) % 3;
        C.I[j] = qgather(L, o + 6, sym_index(0, c0), sym_index(1, c1), sym_index(2, c2));
        C.M[j] = qgather(L, o + 21, sym_index(0, c0), sym_index(1, c1), sym_index(2, c2));
        C.H[j] = qgather(L, o + 12, 0 * 3 + c0, 1 * 3 + c1, 2 * 3 + c2);
        C.Ht[j] = qgather(L, o + 12, c0 * 3 + 0, c1 * 3 + 1, c2 * 3 + 2);  // H^T[r][c] = H[c][r]
    }
}

// replicated (one-lane layout) copy of a 6x6 kept in rotated rows, and of a distributed spatial vector
MPPI_HD AI qai_gather(const QAI &X) {
    AI A;
    // entry (r, c) sits in lane r at rotation j = (c - r + 3) % 3
    A.I = {qget<0>(X.I[0]), qget<0>(X.I[1]), qget<0>(X.I[2]), qget<1>(X.I[0]), qget<1>(X.I[1]), qget<2>(X.I[0])};
    A.M = {qget<0>(X.M[0]), qget<0>(X.M[1]), qget<0>(X.M[2]), qget<1>(X.M[0]), qget<1>(X.M[1]), qget<2>(X.M[0])};
    A.H[0] = qget<0>(X.H[0]); A.H[1] = qget<0>(X.H[1]); A.H[2] = qget<0>(X.H[2]);
    A.H[3] = qget<1>(X.H[2]); A.H[4] = qget<1>(X.H[0]); A.H[5] = qget<1>(X.H[1]);
    A.H[6] = qget<2>(X.H[1]); A.H[7] = qget<2>(X.H[2]); A.H[8] = qget<2>(X.H[0]);
    return A;
}
MPPI_HD V3 qv3_gather(QF x) { return V3{qget<0>(x), qget<1>(x), qget<2>(x)}; }

// rigid inertia about the world origin and velocity-product force of a body posed at (R, p) moving with v:
// the block of quad_aba, shared by the bodies and the floating base
// (h and Tr = R Ic come from the caller: qmoments() for the bodies, the plain sums below for the base)
MPPI_HD void qrigid_tail(const QM3 &R, float mass, QF h, const QF *Tr, const QSV &v, QAI &A, QSV &pA) {
    const float invm = mass > 0.f ? frcp(mass) : 0.f;
    const QF cw = invm * h;
    const QF h1 = rot1(h), h2 = rot2(h);
#if defined(MPPI_DPP_FMAC)
    qinertia_rows_fused(Tr, R.c, h, cw, A.I[0], A.I[1], A.I[2]);   // (rotations folded into the multiply-adds, mppi_quad.hpp)
#else
    const QF hh = qsum(h * cw);
    A.I[0] = Tr[0] * R.c[0] + Tr[1] * R.c[1] + Tr[2] * R.c[2] + hh - h * cw;
    A.I[1] = Tr[0] * rot1(R.c[0]) + Tr[1] * rot1(R.c[1]) + Tr[2] * rot1(R.c[2]) - h * rot1(cw);
    A.I[2] = Tr[0] * rot2(R.c[0]) + Tr[1] * rot2(R.c[1]) + Tr[2] * rot2(R.c[2]) - h * rot2(cw);
#endif
    A.H[0] = qrep(0.f); A.H[1] = -h2;     A.H[2] = h1;   // skew(h), rotated rows
    A.Ht[0] = A.H[0];  A.Ht[1] = h2;      A.Ht[2] = -h1;
    A.M[0] = qrep(mass); A.M[1] = A.H[0]; A.M[2] = A.H[0];
    const QF w = v.a, vl = v.l;
#if defined(MPPI_DPP_FMAC)
    qbias_force_fused(A.I[0], A.I[1], A.I[2], h, qrep(mass), w, vl, pA.a, pA.l);
#else
    const QF n = A.I[0] * w + A.I[1] * rot1(w) + A.I[2] * rot2(w) + qcross(h, vl);
    const QF f = mass * vl + qcross(w, h);
    pA = {qcross(w, n) + qcross(vl, f), qcross(w, f)};
#endif
}

template <class F>
MPPI_HD void qrigid_world(const QM3 &R, QF p, float mass, const F *hb, const F *Ic, const QSV &v, QAI &A, QSV &pA, QF &h) {
    h = R.c[0] * hb[0] + R.c[1] * hb[1] + R.c[2] * hb[2] + mass * p;
    QF Tr[3];
    Tr[0] = R.c[0] * Ic[0] + R.c[1] * Ic[1] + R.c[2] * Ic[2];
    Tr[1] = R.c[0] * Ic[1] + R.c[1] * Ic[3] + R.c[2] * Ic[4];
    Tr[2] = R.c[0] * Ic[2] + R.c[1] * Ic[4] + R.c[2] * Ic[5];
    qrigid_tail(R, mass, h, Tr, v, A, pA);
}

// gravity, contact wrench and implicit contact damping of one frame: (IA + h C) a + (pA + C v - f - f_g) = 0
MPPI_HD void qexternal(const LMem &L, int acc_base, int ent, bool touched, float hstep, QF hmom, float mass, QF gq, const QSV &v, QAI &A,
                       QSV &pA) {
    pA.a = pA.a - qcross(hmom, gq);
    pA.l = pA.l - mass * gq;
    if (touched) {
        QSV fe;
        QAI C;
        qacc_load(L, acc_base, ent, fe, C);
        const QSV Cv = qmul(C, v);
        pA = {pA.a + Cv.a - fe.a, pA.l + Cv.l - fe.l};
        for (int j = 0; j < 3; j++) {
            A.I[j] += hstep * C.I[j];
            A.H[j] += hstep * C.H[j];
            A.Ht[j] += hstep * C.Ht[j];
            A.M[j] += hstep * C.M[j];
        }
    }
}

// Articulated-body solve of the robot inside a contact scene, in two parts so that the second solve of a substep (joint
// drives saturated at their effort limit: other tau / kdh, everything else the same) reuses what does not depend on the
// drive: quad_aba_prepare = velocities, bias terms and the world-frame inertia + bias force of every body incl. gravity,
// contact wrench and implicit contact damping; quad_aba_solve = inward articulated-inertia pass and outward
// accelerations.  vbase / abase: spatial velocity / acceleration of the base about the world origin (zero / unused
// for a fixed base).  `touched`: frames with non-zero accumulators.
template <class T>
struct QAbaPrep {
    static constexpr int NBs = T::NB ? T::NB : 1;
    QSV cb[NBs], pA[NBs + 1];
    QF Sl[NBs];
    QAI A[NBs + 1];  // [NB] = the floating base
};
template <class T, class M>
MPPI_HD void quad_aba_prepare(M &m, const QPose<T> &P, const QSV &vbase, const QF *qd, const LMem &L, unsigned touched, QAbaPrep<T> &W,
                              JointLimits *lim) {
    constexpr int NB = T::NB;
    constexpr int NBs = NB ? NB : 1;
    using Lay = SceneLayout<T>;
    QSV v[NBs];
    const QF zero = qrep(0.f);
    const float hstep = m.h;
    const QF gq = m.gravity_on ? qsel(m.g[0], m.g[1], m.g[2]) : zero;
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const QSV S = quad_subspace<T, i>(P);
        W.Sl[i] = S.l;
        const QSV sj = {qd[i] * S.a, qd[i] * S.l};
        const QSV vp = par < 0 ? vbase : v[par < 0 ? 0 : par];
#if defined(MPPI_DPP_FMAC)
        qvel_bias_fused(vp.a, vp.l, S.a, S.l, qd[i], v[i].a, v[i].l, W.cb[i].a, W.cb[i].l);
#else
        v[i] = {vp.a + sj.a, vp.l + sj.l};
        W.cb[i] = {qcross(vp.a, sj.a), qcross(vp.a, sj.l) + qcross(vp.l, sj.a)};
#endif
    });
    BodyK1 blk[NBs];
    static_rfor<0, NB>([&](auto ic) MPPI_LAMBDA { blk[ic] = load_block<BodyK1>(m.b[ic].k1); });
    static_rfor<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        const BodyK1 &b = blk[i];
        lim[i] = {b.effort, b.vmax};
        QF h, Tr[3];
        qmoments(P.R01[i], P.R2p[i], b, h, Tr);
        qrigid_tail(P.rot(i), b.m, h, Tr, v[i], W.A[i], W.pA[i]);
        qexternal(L, Lay::kAcc, i, (touched >> i) & 1u, hstep, h, b.m, gq, v[i], W.A[i], W.pA[i]);
    });
    if (m.floating != 0) {
        QF h;
        qrigid_world(P.rot_base(), P.pos_base(), m.base_m, m.base_hb, m.base_Ic, vbase, W.A[NB], W.pA[NB], h);
        qexternal(L, Lay::kAcc, NB, (touched >> NB) & 1u, hstep, h, m.base_m, gq, vbase, W.A[NB], W.pA[NB]);
    }
}
template <class T, class M>
MPPI_HD void quad_aba_solve(M &m, const QPose<T> &P, const QAbaPrep<T> &W, const QF *tau_exp, const QF *kdh, QF *qdd, SV &abase) {
    constexpr int NB = T::NB;
    constexpr int NBs = NB ? NB : 1;
    QSV Wn[NBs], pacc[NBs + 1];  // Wn = -U/d and kk = (u - U.c)/d: what the outward pass needs of a joint (qdd = kk + Wn . a_parent)
    QAI acc[NBs + 1];
    QF kk[NBs];
    bool has_acc[NBs + 1];
    const QF zero = qrep(0.f);
    const bool floating = m.floating != 0;
    static_for<0, NB + 1>([&](auto ic) MPPI_LAMBDA { has_acc[ic] = false; });
    static_rfor<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const QSV S = {P.revolute(i) ? P.R2p[i].x : zero, W.Sl[i]};
        QAI A = W.A[i];
        QSV pA = W.pA[i];
        if (has_acc[i]) {
            for (int j = 0; j < 3; j++) { A.I[j] += acc[i].I[j]; A.H[j] += acc[i].H[j]; A.Ht[j] += acc[i].Ht[j]; A.M[j] += acc[i].M[j]; }
            pA = {pA.a + pacc[i].a, pA.l + pacc[i].l};
        }
        const QSV Ui = qmul(A, S);
#if defined(MPPI_DPP_FMAC)
        QF ui, invd;
        qjoint_fused(S.a, S.l, Ui.a, Ui.l, pA.a, pA.l, kdh[i], tau_exp[i], ui, invd, Wn[i].a, Wn[i].l);
#else
        const QF invd = qrcp(qdot6(S, Ui) + kdh[i]), ui = tau_exp[i] - qdot6(S, pA);
        const QF ninvd = -invd;
        Wn[i] = {Ui.a * ninvd, Ui.l * ninvd};
#endif
        const QSV c = W.cb[i];
        constexpr int pj = par < 0 ? NB : par;  // the base accumulator lives at index NB
        if (par >= 0 || floating) {
            const QSV Ac = qmul(A, c);
#if defined(MPPI_DPP_FMAC)
            QSV pa;
            qbias_to_parent_fused(Ui.a, Ui.l, c.a, c.l, ui, invd, pA.a, pA.l, Ac.a, Ac.l, kk[i], pa.a, pa.l);
#else
            const QF k = (ui - qdot6(Ui, c)) * invd;
            kk[i] = k;
            const QSV pa = {pA.a + Ac.a + k * Ui.a, pA.l + Ac.l + k * Ui.l};
#endif
#if defined(MPPI_DPP_FMAC)
            qrank1_fused(A.I, A.H, A.Ht, A.M, Wn[i].a, Wn[i].l, Ui.a, Ui.l);
#else
            const QF un = -Wn[i].a, uf = -Wn[i].l;
            const QF n1 = rot1(Ui.a), n2 = rot2(Ui.a), f1 = rot1(Ui.l), f2 = rot2(Ui.l);
            A.I[0] -= un * Ui.a; A.I[1] -= un * n1; A.I[2] -= un * n2;
            A.H[0] -= un * Ui.l; A.H[1] -= un * f1; A.H[2] -= un * f2;
            A.Ht[0] -= uf * Ui.a; A.Ht[1] -= uf * n1; A.Ht[2] -= uf * n2;
            A.M[0] -= uf * Ui.l; A.M[1] -= uf * f1; A.M[2] -= uf * f2;
#endif
            if (has_acc[pj]) {
                for (int j = 0; j < 3; j++) { acc[pj].I[j] += A.I[j]; acc[pj].H[j] += A.H[j]; acc[pj].Ht[j] += A.Ht[j]; acc[pj].M[j] += A.M[j]; }
                pacc[pj] = {pacc[pj].a + pa.a, pacc[pj].l + pa.l};
            } else {
                acc[pj] = A;
                pacc[pj] = pa;
                has_acc[pj] = true;
            }
        } else {
            kk[i] = (ui - qdot6(Ui, c)) * invd;  // (a fixed root: nothing goes on to a parent)
        }
    });
    abase = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    QSV a0 = {zero, zero};
    if (floating) {
        QAI A = W.A[NB];
        QSV pA = W.pA[NB];
        if (has_acc[NB]) {
            for (int j = 0; j < 3; j++) { A.I[j] += acc[NB].I[j]; A.H[j] += acc[NB].H[j]; A.Ht[j] += acc[NB].Ht[j]; A.M[j] += acc[NB].M[j]; }
            pA = {pA.a + pacc[NB].a, pA.l + pacc[NB].l};
        }
        // the base's 6x6 system, gathered to every lane and solved by the replicated Cholesky
        const V3 na = qv3_gather(pA.a), nl = qv3_gather(pA.l);
        abase = solve6(qai_gather(A), SV{{-na.x, -na.y, -na.z}, {-nl.x, -nl.y, -nl.z}});
        a0 = {qsel(abase.a.x, abase.a.y, abase.a.z), qsel(abase.l.x, abase.l.y, abase.l.z)};
    }
    QSV a[NBs];
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const QSV S = {P.revolute(i) ? P.R2p[i].x : zero, W.Sl[i]};
        const QSV apar = par < 0 ? a0 : a[par < 0 ? 0 : par];
#if defined(MPPI_DPP_FMAC)
        qoutward_fused(Wn[i].a, Wn[i].l, apar.a, apar.l, W.cb[i].a, W.cb[i].l, S.a, S.l, kk[i], qdd[i], a[i].a, a[i].l);
#else
        const QSV ap = {apar.a + W.cb[i].a, apar.l + W.cb[i].l};
        const QF dd = kk[i] + qdot6(Wn[i], apar);
        qdd[i] = dd;
        a[i] = {ap.a + dd * S.a, ap.l + dd * S.l};
#endif
    });
}

// base pose of the sample (its own root row) and the robot's kinematics, distributed over the quad
template <class T, class M>
MPPI_HD void quad_scene_pose(M &m, const SceneState<T> &s, QPose<T> &P) {
    const M3 Rb = quat_to_R(s.base + 3);
    QM3 Rq;
    for (int c = 0; c < 3; c++) Rq.c[c] = qsel(Rb.a[c], Rb.a[3 + c], Rb.a[6 + c]);
    P.set_base(Rq, qsel(s.base[0], s.base[1], s.base[2]));
    QF q[T::NB ? T::NB : 1];
    static_for<0, T::NB>([&](auto ic) MPPI_LAMBDA { q[ic] = qrep(s.q[ic]); });
    quad_fk<T>(m, q, P);
}

// One simulator step of a contact scene, quad layout for the robot (same physics as step_scene).
// `mr0`: the same model as seen by the robot algebra (bodies, links, header) - the kernel passes an LDS copy of that prefix,
// whose blocks arrive in order as VGPRs; shapes, pairs and free bodies keep going through the scalar cache (`m0`).
template <class T, int SPLIT, class M, class MR>
MPPI_HD void step_scene_quad(M &m0, MR &mr0, const float *root, SceneState<T> &s, const float *target, const LMem &L, Split split) {
    constexpr int NB = T::NB;
    constexpr int NBs = NB ? NB : 1;
    M *mp = &m0;
    MR *mrp = &mr0;
    // position mode (reference isaacgym_wrapper.py:571-572): apply_robot_cmd overwrites the DOF state with the command
    if (m0.drive_mode == kDrivePosition)
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA { s.q[ic] = target[ic]; s.qd[ic] = 0.f; });
    for (int sub = 0; sub < m0.substeps; sub++) {
        M &m = *launder(mp);
        MR &mr = *launder(mrp);
        const float h = m.h, kd = m.kd, inv_h = frcp(h);
        QPose<T> P;
        quad_scene_pose<T>(mr, s, P);
        QSV vbase = {qrep(0.f), qrep(0.f)};
        if (m.floating) {
            const QF wb = qsel(s.base[10], s.base[11], s.base[12]), vb = qsel(s.base[7], s.base[8], s.base[9]);
            vbase = {wb, vb - qcross(wb, P.pos_base())};
        }
        QF qd[NBs];
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA { qd[ic] = qrep(s.qd[ic]); });
        {  // dynamic frames into the sample's LDS rows: each lane writes its own matrix rows / vector components
            QSV v[NBs];
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
                constexpr int i = ic;
                constexpr int par = T::par[i];
                const QSV S = quad_subspace<T, i>(P);
                const QSV vp = par < 0 ? vbase : v[par < 0 ? 0 : par];
                v[i] = {vp.a + qd[i] * S.a, vp.l + qd[i] * S.l};
                qframe_store(L, i, P.rot(i), P.pos(i), v[i]);
            });
            qframe_store(L, NB, P.rot_base(), P.pos_base(), vbase);
            for (int f = 0; f < kFreeSlots; f++)
                if (f < m.n_free) {
                    const float *rs = s.fr[f];
                    V3 p = loadv(rs), w = loadv(rs + 10), vl = loadv(rs + 7);
                    frame_store(L, NB + 1 + f, quat_to_R(rs + 3), p, SV{w, vl - cross(w, p)});
                }
        }
        MPPI_SEC(0);
        const unsigned touched = contact_forces<T, SPLIT>(m, root, L, s.acc_dirty, s.cf_dirty, split);
        QF tau[NBs], kdh[NBs], qdd[NBs];
        JointLimits lim[NBs];
        // joint drives and the effort-limit test: as in quad_step (mppi_quad.hpp) - one uniform branch for the drive mode,
        // one running maximum for the saturation test, the selects inside the rare branch
        float kde = kd;
        if (m.drive_mode == kDriveVelocity) {
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA { tau[ic] = kd * (qrep(target[ic]) - qd[ic]); });
        } else if (m.drive_mode == kDriveEffort) {
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA { tau[ic] = qrep(target[ic]) - kd * qd[ic]; });
        } else {  // position: the spring at the end-of-substep position (mppi_quad.hpp quad_step)
            const float kp = m.kp;
            kde = kd + h * kp;
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA { tau[ic] = kp * (qrep(target[ic]) - qrep(s.q[ic])) - kde * qd[ic]; });
        }
        const QF kdhq = qrep(kde * h);
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA { kdh[ic] = kdhq; });
        SV abase;
        QAbaPrep<T> prep;
        quad_aba_prepare<T>(mr, P, vbase, qd, L, touched, prep, lim);
        MPPI_SEC(4);
        quad_aba_solve<T>(mr, P, prep, tau, kdh, qdd, abase);
        MPPI_SEC(5);
        QF tt[NBs];
        QF excess = qrep(-INFINITY);
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            tt[i] = tau[i] - kdhq * qdd[i];
            excess = qmax(excess, qabs(tt[i]) - qrep(lim[i].effort));  // (no limit: effort = +inf)
        });
        if (qany_gt(excess, qrep(0.f))) {
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
                constexpr int i = ic;
                const QF eff = qrep(lim[i].effort);
                const bool sat = qany_gt(qabs(tt[i]), eff);
                tau[i] = sat ? qwhere_gt(tt[i], qrep(0.f), eff, -eff) : tau[i];
                kdh[i] = sat ? qrep(0.f) : kdh[i];
            });
            quad_aba_solve<T>(*launder(mrp), P, prep, tau, kdh, qdd, abase);
        }
        MPPI_SEC(6);
        float dqd[NB ? NB : 1];   // (rate changes of the substep: step_free_bodies)
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            const JointLimits b = lim[i];
            QF v = qd[i] + h * qdd[i];
            // velocity limit and inelastic stops as one pair of bounds (see quad_step in mppi_quad.hpp / joint_limit)
            const QF lo = qrep(mr.b[i].k0.lower), hi = qrep(mr.b[i].k0.upper), z = qrep(0.f), x0 = qrep(s.q[i]);
            const QF vlo = qclamp((lo - x0) * inv_h, qrep(-b.vmax), z), vhi = qclamp((hi - x0) * inv_h, z, qrep(b.vmax));
            v = qclamp(v, vlo, vhi);
            const QF x = qclamp(x0 + h * v, lo, hi);
            s.q[i] = qlane0(x);
            const float vn = qlane0(v);
            dqd[i] = vn - s.qd[i];
            s.qd[i] = vn;
        });
        if (m.floating) root_integrate(s.base, abase, h);
        step_free_bodies<T>(mr, s, L, h, dqd);
        MPPI_SEC(7);
    }
}

// Stage cost of the state after a step and the rollout-visualisation point, both from ONE quad-layout kinematics pass
// (the one-lane path runs a full forward kinematics for each of them).
template <class T, class M, class MR>
MPPI_HD float step_tail_scene_quad(M &m, MR &mr, CCfg &cfg, CCost &c, const float *root, const SceneState<T> &s, const LMem &L, float *viz, int t, int k,
                                   bool leader) {
    const bool want_viz = cfg.want_rollouts && viz != nullptr;
    const bool link_cost = c.kind == kCostBoxerPush || c.kind == kCostPandaPick;
    const bool program = c.kind == kCostProgram;
    float cost;
    QPose<T> P;
    if (link_cost || want_viz || program) quad_scene_pose<T>(mr, s, P);
    if (program) {  // link poses of a cost program from the quad-layout kinematics (gathered to replicated values)
        cost = program_cost_with<T>(c, s.q, s.qd, [&](int l, M3 &R, V3 &p) MPPI_LAMBDA {
            QM3 Rq;
            QF pq;
            quad_link_pose<T>(mr, P, l, Rq, pq);
            for (int cc = 0; cc < 3; cc++) {
                R.a[cc] = qget<0>(Rq.c[cc]);
                R.a[3 + cc] = qget<1>(Rq.c[cc]);
                R.a[6 + cc] = qget<2>(Rq.c[cc]);
            }
            p = qv3_gather(pq);
        }, SceneEnv<T, M>{m, root, s, L});
    } else if (link_cost) {
        QM3 Rq;
        QF pq;
        quad_link_pose<T>(mr, P, c.link[0], Rq, pq);
        M3 R;
        for (int cc = 0; cc < 3; cc++) {
            R.a[cc] = qget<0>(Rq.c[cc]);
            R.a[3 + cc] = qget<1>(Rq.c[cc]);
            R.a[6 + cc] = qget<2>(Rq.c[cc]);
        }
        cost = stage_cost_scene_link<T>(m, c, root, s, L, R, qv3_gather(pq));
    } else {
        cost = stage_cost_scene<T>(m, c, root, s, L);
    }
    if (want_viz) {
        QM3 Rq;
        QF pq;
        quad_link_pose<T>(mr, P, cfg.viz_link, Rq, pq);
        const V3 p = qv3_gather(pq);
        if (leader) {
            const unsigned K = (unsigned)cfg.K;
            viz[(unsigned)(t * 3 + 0) * K + (unsigned)k] = p.x + L.ox;
            viz[(unsigned)(t * 3 + 1) * K + (unsigned)k] = p.y + L.oy;
            viz[(unsigned)(t * 3 + 2) * K + (unsigned)k] = p.z;
        }
    }
    return cost;
}

}  // namespace mppi
