// mppi_scene_oct.hpp - contact-scene step with the articulated-body solve in the OCTET layout (mppi_oct.hpp): the angular half of
// every spatial quantity in one quad of the sample's octet, the linear half in the other.  For the FIXED-BASE trees of more than
// four bodies in a contact scene (the gripper arm on the table, BASELINE configs[4]): the quad-layout solve of
// mppi_scene_quad.hpp is computed twice over, once per quad of the octet, and keeps 17 registers per body alive between its prepare
// and solve halves (QAbaPrep: the kernel needs 450 of the 512); here the two quads share the work and the prepared state is
// 9 registers per body.  Floating-base trees keep the quad-layout path (their 6x6 base system is solved replicated).
//
// Same physics as step_scene_quad / step_scene (explicit gravity, contact wrench and implicit contact damping from the sample's
// LDS accumulators, drives, effort saturation with one re-solve, joint limits); device only.
#pragma once
#include "mppi_oct.hpp"
#include "mppi_scene_quad.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
namespace mppi {

// this lane's entries of a frame's accumulator block [27] = f(6) | I(6 sym) | H(9) | M(6 sym) in the octet layout: the external
// wrench component, the own-half rows D (I | M) and the cross rows O (H | H^T), rotated rows x[j] = X[r][(r+j)%3]
struct OctAccOff {
    int fe, d[3], o[3];
};
__device__ __forceinline__ OctAccOff oct_acc_offsets() {
    const int r = quad_row(), half = oct_half();
    OctAccOff f;
    f.fe = 3 * half + r;
    for (int j = 0; j < 3; j++) {
        const int c = (r + j) % 3, lo = r < c ? r : c, hi = r < c ? c : r;
        f.d[j] = (half ? 21 : 6) + lo * 3 - lo * (lo - 1) / 2 + (hi - lo);
        f.o[j] = 12 + (half ? 3 * c + r : 3 * r + c);   // H[r][c] | H^T[r][c] = H[c][r]
    }
    return f;
}
// frame row block of mppi_scene.hpp: R[9] row-major, p[3], w[3], vO[3] - the rotation and position by every lane's own rows (both
// halves write the same values), the velocity by one store: angular lanes w_r, linear lanes vO_r
__device__ __forceinline__ void oframe_store(const LMem &L, int ent, const QM3 &R, QF p, OF v) {
    const int o = ent * 18;
    for (int c = 0; c < 3; c++) qscatter(L, o + c, 0, 3, 6, R.c[c]);
    qscatter(L, o + 9, 0, 1, 2, p);
    qscatter(L, o + 12 + 3 * oct_half(), 0, 1, 2, v);
}

// pass 1 for the frames: S, v = vp + qd S, w = wp + qd az of one body (the bias term comes with the solve's own pass, after the
// candidate pairs have been walked - nothing of this survives them in registers)
__device__ __forceinline__ void opass1_vel_fused(OF p, OF az, OF qd, OF vp, OF wp, OF &v, OF &w) {
    OF S, saz, t, sj;
    asm("v_mov_b32 %[S], %[az]\n\t"                               //  1 S   = az
        "v_mul_f32 %[saz], %[qd], %[az]\n\t"                      //  2 saz = qd az
        "v_mul_f32_dpp %[t], %[az], %[p] " MPPI_R1 "\n\t"         //  3 t   = rot1(az) p
        "v_fmac_f32_dpp %[t], %[p], -%[az] " MPPI_R1 "\n\t"       //  4 t  -= rot1(p) az
        "v_add_f32 %[w], %[wp], %[saz]\n\t"                       //  5 w   = wp + saz
        "s_nop 0\n\t"                                             //  6
        "v_mov_b32_dpp %[S], %[t] quad_perm:[1,2,0,1] row_mask:0xf bank_mask:0xc\n\t"   //  7 S (linear lanes) = rot1(t)     (t written at 4)
        "v_mul_f32 %[sj], %[qd], %[S]\n\t"                        //  8 sj  = qd S
        "v_add_f32 %[v], %[vp], %[sj]"                            //  9 v   = vp + sj
        : [v] "=&v"(v), [w] "=&v"(w), [S] "=&v"(S), [saz] "=&v"(saz), [t] "=&v"(t), [sj] "=&v"(sj)
        : [p] "v"(p), [az] "v"(az), [qd] "v"(qd), [vp] "v"(vp), [wp] "v"(wp));
}

// rigid inertia rows, cross rows and bias force of one body incl. gravity (prepare half of the solve: nothing here depends on the
// joint drives).  gA = -ang g_{r+1}, gB = ang g_r: the moment h x g before its rotation; gL = -lin g_r: the force m g
struct OctGravity {
    OF gA, gB, gL;
};
__device__ __forceinline__ void oprep_fused(const OctBodyIn &b, const OctLane &ol, const OctGravity &G, OAI &A, OF &pA) {
    OF cw, t, s, g, th, A2, gs, t3;
    asm(MPPI_OCT_INERTIA MPPI_OCT_G
        "v_mul_f32_dpp %[O1], %[h], %[nsg] " MPPI_R2 "\n\t"       // 21 O1  = rot2(h) nsg
        "v_fmac_f32_dpp %[g], %[th], %[sg] " MPPI_R1 "\n\t"       // 22 g  += rot1(th) sg         (th written at 19)
        "v_mul_f32_dpp %[O2], %[h], %[sg] " MPPI_R1 "\n\t"        // 23 O2  = rot1(h) sg
        "v_mul_f32 %[t3], %[h], %[gA]\n\t"                        // 24 t3  = h gA                (gravity moment, before its rotation)
        "v_mov_b32_dpp %[gs], %[g] " MPPI_SW "\n\t"               // 25 gs  = swap(g)             (g written at 22)
        "v_fmac_f32_dpp %[t3], %[h], %[gB] " MPPI_R1 "\n\t"       // 26 t3 += rot1(h) gB
        "v_fmac_f32_dpp %[t3], %[g], %[w] " MPPI_R1 "\n\t"        // 27 t3 += rot1(g) w
        "v_fmac_f32_dpp %[t3], %[w], -%[g] " MPPI_R1 "\n\t"       // 28 t3 -= rot1(w) g
        "v_fmac_f32_dpp %[t3], %[gs], %[A2] " MPPI_R1 "\n\t"      // 29 t3 += rot1(gs) A2         (gs written at 25)
        "v_fmac_f32_dpp %[t3], %[A2], -%[gs] " MPPI_R1 "\n\t"     // 30 t3 -= rot1(A2) gs
        "s_nop 1\n\t"                                             // 31                           (t3 written at 30)
        "v_mov_b32_dpp %[pA], %[t3] " MPPI_R1 "\n\t"              // 32 pA  = rot1(t3)
        "v_fmac_f32 %[pA], %[m], %[gL]"                           // 33 pA -= m g in the linear lanes
        : [D0] "=&v"(A.D[0]), [D1] "=&v"(A.D[1]), [D2] "=&v"(A.D[2]), [O1] "=&v"(A.O[1]), [O2] "=&v"(A.O[2]), [pA] "=&v"(pA), [cw] "=&v"(cw), [t] "=&v"(t),
          [s] "=&v"(s), [g] "=&v"(g), [th] "=&v"(th), [A2] "=&v"(A2), [gs] "=&v"(gs), [t3] "=&v"(t3)
        : [Tr0] "v"(b.Tr0), [Tr1] "v"(b.Tr1), [Tr2] "v"(b.Tr2), [h] "v"(b.h), [R0] "v"(b.R0), [R1] "v"(b.R1), [R2] "v"(b.R2), [m] "v"(b.m), [invm] "v"(b.invm),
          [v] "v"(b.v), [w] "v"(b.w), [sg] "v"(ol.sg), [nsg] "v"(ol.nsg), [ang] "v"(ol.ang), [lin] "v"(ol.lin), [gA] "v"(G.gA), [gB] "v"(G.gB), [gL] "v"(G.gL));
    A.O[0] = 0.f;
}
// y = C x for the implicit contact damping of a touched frame (x = the frame's velocity)
__device__ __forceinline__ OF omul_fused(const OAI &A, OF x) {
    OF X, sx;
    asm("v_mul_f32 %[X], %[D0], %[x]\n\t"                         //  1 X  = D0 x
        "s_nop 0\n\t"                                             //  2                     (x: maybe written right in front of the block)
        "v_mov_b32_dpp %[sx], %[x] " MPPI_SW "\n\t"               //  3 sx = swap(x)
        "v_fmac_f32_dpp %[X], %[x], %[D1] " MPPI_R1 "\n\t"        //  4 X += rot1(x) D1
        "v_fmac_f32_dpp %[X], %[x], %[D2] " MPPI_R2 "\n\t"        //  5 X += rot2(x) D2
        "v_fmac_f32 %[X], %[O0], %[sx]\n\t"                       //  6 X += O0 sx
        "v_fmac_f32_dpp %[X], %[sx], %[O1] " MPPI_R1 "\n\t"       //  7 X += rot1(sx) O1    (sx written at 3)
        "v_fmac_f32_dpp %[X], %[sx], %[O2] " MPPI_R2              //  8 X += rot2(sx) O2
        : [X] "=&v"(X), [sx] "=&v"(sx)
        : [D0] "v"(A.D[0]), [D1] "v"(A.D[1]), [D2] "v"(A.D[2]), [O0] "v"(A.O[0]), [O1] "v"(A.O[1]), [O2] "v"(A.O[2]), [x] "v"(x));
    return X;
}
// solve half of one body: (U, V) = IA (S, c), the joint, Ia = IA + W rot(U) and pa for the parent.  A: this body's inertia incl. its
// children's (modified in place), pA likewise
__device__ __forceinline__ void osolve_fused(OAI &A, OF pA, OF S, OF cb, OF kdh, OF tau, OF &W, OF &k, OF &pa) {
    OF sx, sy, X, Y, t1, t2, tk, d, u, invd, sU;
    asm("v_mul_f32 %[X], %[D0], %[S]\n\t"                         //  1 X   = D0 S
        "v_mul_f32 %[Y], %[D0], %[cb]\n\t"                        //  2 Y   = D0 c
        "v_mov_b32_dpp %[sx], %[S] " MPPI_SW "\n\t"               //  3 sx  = swap(S)
        "v_mov_b32_dpp %[sy], %[cb] " MPPI_SW "\n\t"              //  4 sy  = swap(c)
        "v_fmac_f32_dpp %[X], %[S], %[D1] " MPPI_R1 "\n\t"        //  5 X  += rot1(S) D1
        "v_fmac_f32_dpp %[Y], %[cb], %[D1] " MPPI_R1 "\n\t"       //  6 Y  += rot1(c) D1
        "v_fmac_f32_dpp %[X], %[S], %[D2] " MPPI_R2 "\n\t"        //  7 X  += rot2(S) D2
        "v_fmac_f32_dpp %[Y], %[cb], %[D2] " MPPI_R2 "\n\t"       //  8 Y  += rot2(c) D2
        "v_fmac_f32 %[X], %[O0], %[sx]\n\t"                       //  9 X  += O0 sx
        "v_fmac_f32 %[Y], %[O0], %[sy]\n\t"                       // 10 Y  += O0 sy
        "v_fmac_f32_dpp %[X], %[sx], %[O1] " MPPI_R1 "\n\t"       // 11 X  += rot1(sx) O1         (sx written at 3)
        "v_fmac_f32_dpp %[Y], %[sy], %[O1] " MPPI_R1 "\n\t"       // 12 Y  += rot1(sy) O1
        "v_fmac_f32_dpp %[X], %[sx], %[O2] " MPPI_R2 "\n\t"       // 13 X  += rot2(sx) O2
        "v_fmac_f32_dpp %[Y], %[sy], %[O2] " MPPI_R2 "\n\t"       // 14 Y  += rot2(sy) O2
        MPPI_OCT_JOINT                                            // 15-38
        "v_fmac_f32 %[O0], %[W], %[sU]\n\t"                       // 39 O0 += W sU                (sU written at 35)
        "v_fmac_f32_dpp %[O1], %[sU], %[W] " MPPI_R1 "\n\t"       // 40 O1 += rot1(sU) W
        "v_fmac_f32_dpp %[O2], %[sU], %[W] " MPPI_R2              // 41 O2 += rot2(sU) W
        : [D0] "+v"(A.D[0]), [D1] "+v"(A.D[1]), [D2] "+v"(A.D[2]), [O0] "+v"(A.O[0]), [O1] "+v"(A.O[1]), [O2] "+v"(A.O[2]), [W] "=&v"(W), [k] "=&v"(k),
          [pa] "=&v"(pa), [sx] "=&v"(sx), [sy] "=&v"(sy), [X] "=&v"(X), [Y] "=&v"(Y), [t1] "=&v"(t1), [t2] "=&v"(t2), [tk] "=&v"(tk), [d] "=&v"(d),
          [u] "=&v"(u), [invd] "=&v"(invd), [sU] "=&v"(sU)
        : [pA] "v"(pA), [S] "v"(S), [cb] "v"(cb), [kdh] "v"(kdh), [tau] "v"(tau));
}
// ... of the root of a fixed-base tree: W = -U/d and k = u/d
__device__ __forceinline__ void osolve_root_fused(const OAI &A, OF pA, OF S, OF kdh, OF tau, OF &W, OF &k) {
    OF sx, X, t1, t2, d, u, invd;
    asm("v_mul_f32 %[X], %[D0], %[S]\n\t"                         //  1 X   = D0 S
        "s_nop 0\n\t"                                             //  2
        "v_mov_b32_dpp %[sx], %[S] " MPPI_SW "\n\t"               //  3 sx  = swap(S)
        "v_fmac_f32_dpp %[X], %[S], %[D1] " MPPI_R1 "\n\t"        //  4 X  += rot1(S) D1
        "v_fmac_f32_dpp %[X], %[S], %[D2] " MPPI_R2 "\n\t"        //  5 X  += rot2(S) D2
        "v_fmac_f32 %[X], %[O0], %[sx]\n\t"                       //  6 X  += O0 sx
        "v_fmac_f32_dpp %[X], %[sx], %[O1] " MPPI_R1 "\n\t"       //  7 X  += rot1(sx) O1         (sx written at 3)
        "v_fmac_f32_dpp %[X], %[sx], %[O2] " MPPI_R2 "\n\t"       //  8 X  += rot2(sx) O2
        MPPI_OCT_JOINT_ROOT
        : [W] "=&v"(W), [k] "=&v"(k), [sx] "=&v"(sx), [X] "=&v"(X), [t1] "=&v"(t1), [t2] "=&v"(t2), [d] "=&v"(d), [u] "=&v"(u), [invd] "=&v"(invd)
        : [D0] "v"(A.D[0]), [D1] "v"(A.D[1]), [D2] "v"(A.D[2]), [O0] "v"(A.O[0]), [O1] "v"(A.O[1]), [O2] "v"(A.O[2]), [pA] "v"(pA), [S] "v"(S), [kdh] "v"(kdh),
          [tau] "v"(tau));
}

// what the solve needs of every body, kept between the first and a second (effort-saturated) solve of a substep
template <class T>
struct OAbaPrep {
    static constexpr int NBs = T::NB ? T::NB : 1;
    OF S[NBs], cb[NBs], pA[NBs];
    OAI A[NBs];
};
template <class T, class M>
__device__ __forceinline__ void oct_scene_prepare(M &m, OctBodies bodies, const OctLane &ol, const QPose<T> &P, const QF *qd, const LMem &L, unsigned touched,
                                                  OAbaPrep<T> &W, JointLimits *lim) {
    constexpr int NB = T::NB;
    constexpr int NBs = NB ? NB : 1;
    using Lay = SceneLayout<T>;
    const OF zero = 0.f;
    const float hstep = m.h;
    OF v[NBs], w[NBs];
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        constexpr int pj = par < 0 ? 0 : par;
        const OF az = P.R2p[i].x;
        if (P.revolute(i)) {
            if constexpr (par < 0) {
                opass1_root_fused(P.pos(i), az, qd[i], W.S[i], v[i], w[i]);
                W.cb[i] = zero;
            } else {
                opass1_fused(P.pos(i), az, qd[i], v[pj], w[pj], ol.lin, W.S[i], v[i], w[i], W.cb[i]);
            }
        } else {
            W.S[i] = ol.lin * az;
            const OF sj = qd[i] * W.S[i];
            if constexpr (par < 0) {
                v[i] = sj;
                w[i] = zero;
                W.cb[i] = zero;
            } else {
                v[i] = v[pj] + sj;
                w[i] = w[pj];
                W.cb[i] = qcross(w[pj], sj);
            }
        }
    });
    OctGravity G{zero, zero, zero};
    if (m.gravity_on) {
        const OF g0 = qsel(m.g[0], m.g[1], m.g[2]), g1 = qsel(m.g[1], m.g[2], m.g[0]);
        G = OctGravity{-ol.ang * g1, ol.ang * g0, -ol.lin * g0};
    }
    const OctAccOff off = oct_acc_offsets();
    BodyK1 blk[NBs];
    static_rfor<0, NB>([&](auto ic) MPPI_LAMBDA { blk[ic] = load_block<BodyK1>(bodies[ic].k1); });
    static_rfor<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        const BodyK1 &b = blk[i];
        lim[i] = {b.effort, b.vmax};
        OF h, Tr[3];
        qmoments(P.R01[i], P.R2p[i], b, h, Tr);
        const OctBodyIn in{Tr[0], Tr[1], Tr[2], h, P.R01[i].x, P.R01[i].y, P.R2p[i].x, b.m, b.invm, v[i], w[i], zero, zero, zero, zero};
        oprep_fused(in, ol, G, W.A[i], W.pA[i]);
        if ((touched >> i) & 1u) {   // contact wrench and implicit contact damping of this frame:  (IA + h C) a + (pA + C v - f) = 0
            const int o = Lay::kAcc + 27 * i;
            OAI C;
            const OF fe = L[o + off.fe];
            for (int j = 0; j < 3; j++) {
                C.D[j] = L[o + off.d[j]];
                C.O[j] = L[o + off.o[j]];
            }
            W.pA[i] += omul_fused(C, v[i]) - fe;
            for (int j = 0; j < 3; j++) {
                W.A[i].D[j] += hstep * C.D[j];
                W.A[i].O[j] += hstep * C.O[j];
            }
        }
    });
}
template <class T, class M>
__device__ __forceinline__ void oct_scene_solve(M &m, const OAbaPrep<T> &W, const QF *tau_exp, const QF *kdh, QF *qdd) {
    constexpr int NB = T::NB;
    constexpr int NBs = NB ? NB : 1;
    OF Wn[NBs], kk[NBs], pacc[NBs];
    OAI acc[NBs];
    bool has_acc[NBs];
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA { has_acc[ic] = false; });
    static_rfor<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        OAI A = W.A[i];
        OF pA = W.pA[i];
        if (has_acc[i]) {
            for (int j = 0; j < 3; j++) { A.D[j] += acc[i].D[j]; A.O[j] += acc[i].O[j]; }
            pA += pacc[i];
        }
        if constexpr (par < 0) {
            osolve_root_fused(A, pA, W.S[i], kdh[i], tau_exp[i], Wn[i], kk[i]);
        } else {
            OF pa;
            osolve_fused(A, pA, W.S[i], W.cb[i], kdh[i], tau_exp[i], Wn[i], kk[i], pa);
            constexpr int pj = par < 0 ? 0 : par;
            if (has_acc[pj]) {
                for (int j = 0; j < 3; j++) { acc[pj].D[j] += A.D[j]; acc[pj].O[j] += A.O[j]; }
                pacc[pj] += pa;
            } else {
                acc[pj] = A;
                pacc[pj] = pa;
                has_acc[pj] = true;
            }
        }
    });
    OF a[NBs];
    const OF zero = 0.f;
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const OF apar = par >= 0 ? a[par < 0 ? 0 : par] : zero;   // (gravity is explicit here: the fixed base does not accelerate)
        ooutward_fused(Wn[i], apar, W.cb[i], W.S[i], kk[i], qdd[i], a[i]);
    });
}

// One simulator step of a contact scene, fixed-base tree, octet layout of the solve (same physics as step_scene_quad)
template <class T, int SPLIT, class M, class MR>
__device__ __forceinline__ void step_scene_oct(M &m0, MR &mr0, const float *root, SceneState<T> &s, const float *target, const LMem &L, Split split) {
    constexpr int NB = T::NB;
    constexpr int NBs = NB ? NB : 1;
    M *mp = &m0;
    MR *mrp = &mr0;
    const OctLane ol = oct_lane();
    OctBodies bodies = (OctBodies)(unsigned long)L.oct_bodies;
    asm volatile("" : "+v"(bodies));
    // position mode (reference isaacgym_wrapper.py:571-572): apply_robot_cmd overwrites the DOF state with the command
    if (m0.drive_mode == kDrivePosition)
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA { s.q[ic] = target[ic]; s.qd[ic] = 0.f; });
    for (int sub = 0; sub < m0.substeps; sub++) {
        M &m = *launder(mp);
        MR &mr = *launder(mrp);
        const float h = m.h, kd = m.kd, inv_h = frcp(h);
        QPose<T> P;
        quad_scene_pose<T>(mr, s, P);
        QF qd[NBs];
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA { qd[ic] = qrep(s.qd[ic]); });
        {  // dynamic frames into the sample's LDS rows
            OF v[NBs], w[NBs];
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
                constexpr int i = ic;
                constexpr int par = T::par[i];
                constexpr int pj = par < 0 ? 0 : par;
                const OF az = P.R2p[i].x;
                if (P.revolute(i)) {
                    opass1_vel_fused(P.pos(i), az, qd[i], par < 0 ? 0.f : v[pj], par < 0 ? 0.f : w[pj], v[i], w[i]);
                } else {
                    const OF sj = qd[i] * (ol.lin * az);
                    v[i] = par < 0 ? sj : v[pj] + sj;
                    w[i] = par < 0 ? 0.f : w[pj];
                }
                oframe_store(L, i, P.rot(i), P.pos(i), v[i]);
            });
            const QSV vb0 = {qrep(0.f), qrep(0.f)};
            qframe_store(L, NB, P.rot_base(), P.pos_base(), vb0);
            for (int f = 0; f < kFreeSlots; f++)
                if (f < m.n_free) {
                    const float *rs = s.fr[f];
                    V3 p = loadv(rs), w3 = loadv(rs + 10), vl = loadv(rs + 7);
                    frame_store(L, NB + 1 + f, quat_to_R(rs + 3), p, SV{w3, vl - cross(w3, p)});
                }
        }
        MPPI_SEC(0);
        const unsigned touched = contact_forces<T, SPLIT>(m, root, L, s.acc_dirty, s.cf_dirty, split);
        QF tau[NBs], kdh[NBs], qdd[NBs];
        JointLimits lim[NBs];
        float kde = kd;
        if (m.drive_mode == kDriveVelocity) {
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA { tau[ic] = kd * (qrep(target[ic]) - qd[ic]); });
        } else if (m.drive_mode == kDriveEffort) {
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA { tau[ic] = qrep(target[ic]) - kd * qd[ic]; });
        } else {
            const float kp = m.kp;
            kde = kd + h * kp;
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA { tau[ic] = kp * (qrep(target[ic]) - qrep(s.q[ic])) - kde * qd[ic]; });
        }
        const QF kdhq = qrep(kde * h);
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA { kdh[ic] = kdhq; });
        OAbaPrep<T> prep;
        oct_scene_prepare<T>(mr, bodies, ol, P, qd, L, touched, prep, lim);
        MPPI_SEC(4);
        oct_scene_solve<T>(mr, prep, tau, kdh, qdd);
        MPPI_SEC(5);
        QF tt[NBs];
        QF excess = qrep(-INFINITY);
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            tt[i] = tau[i] - kdhq * qdd[i];
            excess = qmax(excess, qabs(tt[i]) - qrep(lim[i].effort));
        });
        if (qany_gt(excess, qrep(0.f))) {
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
                constexpr int i = ic;
                const QF eff = qrep(lim[i].effort);
                const bool sat = qany_gt(qabs(tt[i]), eff);
                tau[i] = sat ? qwhere_gt(tt[i], qrep(0.f), eff, -eff) : tau[i];
                kdh[i] = sat ? qrep(0.f) : kdh[i];
            });
            oct_scene_solve<T>(*launder(mrp), prep, tau, kdh, qdd);
        }
        MPPI_SEC(6);
        float dqd[NB ? NB : 1];   // (rate changes of the substep: step_free_bodies)
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            const JointLimits b = lim[i];
            QF v = qd[i] + h * qdd[i];
            const QF lo = qrep(mr.b[i].k0.lower), hi = qrep(mr.b[i].k0.upper), z = qrep(0.f), x0 = qrep(s.q[i]);
            const QF vlo = qclamp((lo - x0) * inv_h, qrep(-b.vmax), z), vhi = qclamp((hi - x0) * inv_h, z, qrep(b.vmax));
            v = qclamp(v, vlo, vhi);
            const QF x = qclamp(x0 + h * v, lo, hi);
            s.q[i] = qlane0(x);
            const float vn = qlane0(v);
            dqd[i] = vn - s.qd[i];
            s.qd[i] = vn;
        });
        step_free_bodies<T>(mr, s, L, h, dqd);
        MPPI_SEC(7);
    }
}

}  // namespace mppi
#endif
