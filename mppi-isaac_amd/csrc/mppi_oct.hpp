// mppi_oct.hpp - articulated-body solve with ONE SAMPLE PER TWO QUADS (8 lanes): the ANGULAR half of every spatial quantity
// lives in one quad, the LINEAR half in the other (fixed-base, contact-free scenes; device only).
//
// Why: k_rollout_quad is bound by the instruction count of its lone wavefronts (one instruction per ~4.75 cycles whatever it
// is, tools/exp/issue_rate.hip).  In the quad layout (mppi_quad.hpp) a spatial vector is TWO registers (a, l) and a 6x6
// articulated inertia twelve; every 6x6 product, rank-one update and spatial sum is issued once for the angular and once for the
// linear rows.  With the halves in different lanes the same instruction serves both: a spatial vector is ONE register, the 6x6
// six (own-half blocks D = I | M, cross blocks O = H | H^T in rotated rows), y = A x is a swap + 6 multiply-adds instead of
// 10 instructions, a rank-one update 7 instead of 10, spatial sums 1 instead of 2.  What does not split - the rigid inertia
// R Ic R^T, the kinematics, scalars replicated over the lanes - is computed by both halves alike.  Counted per body of a chain:
// 106 issue slots against 130 (tools/exp/oct_aba_proto.hip measures both on the same inputs).
//
// Lane map: within every 16-lane row, quads 0 and 1 hold the angular halves of two samples, quads 2 and 3 the linear halves of
// the same two samples.  Partner lanes are 8 apart inside the row, so ONE DPP control - row_ror:8 - swaps the halves of every
// sample in the wavefront (measured lane map: tools/exp/dpp_probe.hip); both quads of a sample hold components 0, 1, 2, 0 in
// the same order, so the quad layout's rotations (quad_perm) mean the same in either half and replicated scalars come out
// bit-identical in all eight lanes (a_r + l_r is commutative; the sum over r runs in the same order everywhere).
// Component layout inside a quad, rotated rows, cross products: exactly mppi_quad.hpp.
//
// Hazards (the compiler does not look into inline assembly; tools/check_dpp_hazards.py checks the built library): a VGPR read
// through DPP needs two wait states after its write - the blocks below order their instructions for that and say where a wait
// is left (s_nop); the result of a transcendental is not read in the next issue slot.  Blocks whose FIRST instruction reads an
// input through DPP start with MPPI_LEAD (mppi_quad.hpp: two wait states in translation units whose kernels reload operands
// from the AGPR file right in front of inline assembly - trees of more than nine bodies; nothing elsewhere).
#pragma once
#include "mppi_quad.hpp"

#if defined(__HIP_DEVICE_COMPILE__)
namespace mppi {

typedef float OF;
// per-lane constants of the half a lane serves
struct OctLane {
    float ang, lin;   // 1 / 0 in the angular lanes, 0 / 1 in the linear ones
    float sg, nsg;    // +1 / -1 and its negative
};
__device__ __forceinline__ int oct_half() { return (int)((threadIdx.x >> 3) & 1u); }
// sample slot (0..7) of this lane inside its wavefront: two samples per 16-lane row
__device__ __forceinline__ int oct_slot() { return (int)(((threadIdx.x >> 4) & 3u) * 2u + ((threadIdx.x >> 2) & 1u)); }
__device__ __forceinline__ OctLane oct_lane() {
    const bool l = oct_half() != 0;
    return OctLane{l ? 0.f : 1.f, l ? 1.f : 0.f, l ? -1.f : 1.f, l ? 1.f : -1.f};
}
__device__ __forceinline__ OF oswap(OF x) { return quad_dpp<0x128>(x); }  // row_ror:8: the other half of the same sample

struct OAI {  // 6x6 [[I, H], [H^T, M]]: angular lanes hold rows of [I H], linear lanes rows of [H^T M]; rotated rows x[j] = X[r][(r+j)%3]
    OF D[3], O[3];
};

#define MPPI_SW "row_ror:8 row_mask:0xf bank_mask:0xf bound_ctrl:1"

// ---- the solve's inline-assembly blocks ------------------------------------------------------------------------------------------
// One block per body and pass: everything a body needs in a pass is ONE asm statement, for two reasons measured on the first
// version of this file (one block per formula, as in mppi_quad.hpp): the compiler puts a wait state between two dependent asm
// statements (its hazard recogniser treats an asm's outputs like a packed instruction's: ~6 s_nop per body), and the waits the
// DPP hazards need inside a short block find no useful instruction to hide behind.  In the long blocks below the slots are
// ordered by hand so that every DPP read sits at least three slots behind the write of its operand; the few waits left are
// marked.  Operands are named; (written at N) notes the slot a DPP-read operand was produced in.
#define MPPI_OCT_INERTIA                                                                                                             \
    "v_mul_f32 %[cw], %[invm], %[h]\n\t"                      /*  1 cw  = h / m            (0 in the linear lanes: their 1/m is 0) */ \
    "v_mul_f32 %[t], %[h], %[cw]\n\t"                         /*  2 t   = h cw                                                    */ \
    "v_mul_f32 %[D0], %[Tr0], %[R0]\n\t"                      /*  3 D0  = Tr0 R0                                                  */ \
    "v_fmac_f32 %[D0], %[Tr1], %[R1]\n\t"                     /*  4 D0 += Tr1 R1                                                  */ \
    "v_fmac_f32 %[D0], %[Tr2], %[R2]\n\t"                     /*  5 D0 += Tr2 R2                                                  */ \
    "v_mul_f32_dpp %[D1], %[R0], %[Tr0] " MPPI_R1 "\n\t"      /*  6 D1  = rot1(R0) Tr0                                            */ \
    "v_fmac_f32_dpp %[D1], %[R1], %[Tr1] " MPPI_R1 "\n\t"     /*  7 D1 += rot1(R1) Tr1                                            */ \
    "v_fmac_f32_dpp %[D1], %[R2], %[Tr2] " MPPI_R1 "\n\t"     /*  8 D1 += rot1(R2) Tr2                                            */ \
    "v_fmac_f32_dpp %[D1], %[cw], -%[h] " MPPI_R1 "\n\t"      /*  9 D1 -= rot1(cw) h          (cw written at 1)                   */ \
    "v_add_f32_dpp %[D0], %[t], %[D0] " MPPI_R1 "\n\t"        /* 10 D0 += rot1(t)             (t written at 2): the diagonal      */ \
    "v_add_f32_dpp %[D0], %[t], %[D0] " MPPI_R2 "\n\t"        /* 11 D0 += rot2(t)             |h|^2/m - h_r c_r = the other two   */ \
    "v_mov_b32_dpp %[D2], %[D1] " MPPI_R2 "\n\t"              /* 12 D2  = rot2(D1)            (D1 written at 9): I is symmetric,  */ \
                                                              /*    I[r][r+2] = I[r+2][r] = the first off-diagonal two rows on    */ \
    "v_fmac_f32 %[D0], %[lin], %[m]\n\t"                      /* 13 D0 += m in the linear lanes (their rows above are zero)       */
#define MPPI_OCT_G                                                                                                                   \
    "v_mov_b32_dpp %[s], %[v] " MPPI_SW "\n\t"                /* 14 s   = swap(v)                                                 */ \
    "v_mul_f32 %[g], %[D0], %[v]\n\t"                         /* 15 g   = D0 v                                                    */ \
    "v_fmac_f32_dpp %[g], %[v], %[D1] " MPPI_R1 "\n\t"        /* 16 g  += rot1(v) D1                                              */ \
    "v_fmac_f32_dpp %[g], %[v], %[D2] " MPPI_R2 "\n\t"        /* 17 g  += rot2(v) D2          (last read of the RIGID rows)       */ \
    "v_mul_f32_dpp %[th], %[s], %[h] " MPPI_R1 "\n\t"         /* 18 th  = rot1(s) h           (s written at 14)                   */ \
    "v_fmac_f32_dpp %[th], %[h], -%[s] " MPPI_R1 "\n\t"       /* 19 th -= rot1(h) s                                               */ \
    "v_mul_f32 %[A2], %[ang], %[s]\n\t"                       /* 20 A2  = vl in the angular lanes, 0 in the linear                */
#define MPPI_OCT_T3                                                                                                                  \
    "v_mul_f32_dpp %[t3], %[g], %[w] " MPPI_R1 "\n\t"         /*    t3  = rot1(g) w                                               */ \
    "v_fmac_f32_dpp %[t3], %[w], -%[g] " MPPI_R1 "\n\t"       /*    t3 -= rot1(w) g                                               */ \
    "v_fmac_f32_dpp %[t3], %[gs], %[A2] " MPPI_R1 "\n\t"      /*    t3 += rot1(gs) A2         (gs written three slots back)       */ \
    "v_fmac_f32_dpp %[t3], %[A2], -%[gs] " MPPI_R1 "\n\t"     /*    t3 -= rot1(A2) gs                                             */
#define MPPI_OCT_JOINT                                                                                                               \
    "v_mul_f32 %[t1], %[S], %[X]\n\t"                         /*    t1  = S U                                                     */ \
    "v_mul_f32 %[t2], %[S], %[pA]\n\t"                        /*    t2  = S pA                                                    */ \
    "v_mul_f32 %[tk], %[X], %[cb]\n\t"                        /*    tk  = U c                                                     */ \
    "v_add_f32_dpp %[t1], %[t1], %[t1] " MPPI_SW "\n\t"       /*    t1 += swap(t1)            (t1 written three slots back)       */ \
    "v_add_f32_dpp %[t2], %[t2], %[t2] " MPPI_SW "\n\t"       /*    t2 += swap(t2)                                                */ \
    "v_add_f32_dpp %[tk], %[tk], %[tk] " MPPI_SW "\n\t"       /*    tk += swap(tk)                                                */ \
    "v_add_f32_dpp %[d], %[t1], %[kdh] " MPPI_B(0) "\n\t"     /*    d   = t1[0] + kdh                                             */ \
    "v_subrev_f32_dpp %[u], %[t2], %[tau] " MPPI_B(0) "\n\t"  /*    u   = tau - t2[0]                                             */ \
    "v_add_f32_dpp %[d], %[t1], %[d] " MPPI_B(1) "\n\t"       /*    d  += t1[1]                                                   */ \
    "v_subrev_f32_dpp %[u], %[t2], %[u] " MPPI_B(1) "\n\t"    /*    u  -= t2[1]                                                   */ \
    "v_add_f32_dpp %[d], %[t1], %[d] " MPPI_B(2) "\n\t"       /*    d  += t1[2]                                                   */ \
    "v_subrev_f32_dpp %[u], %[t2], %[u] " MPPI_B(2) "\n\t"    /*    u  -= t2[2]                                                   */ \
    "v_rcp_f32 %[invd], %[d]\n\t"                             /*    1/d                                                           */ \
    "v_subrev_f32_dpp %[k], %[tk], %[u] " MPPI_B(0) "\n\t"    /*    k   = u - tk[0]                                               */ \
    "v_subrev_f32_dpp %[k], %[tk], %[k] " MPPI_B(1) "\n\t"    /*    k  -= tk[1]                                                   */ \
    "v_subrev_f32_dpp %[k], %[tk], %[k] " MPPI_B(2) "\n\t"    /*    k  -= tk[2]                                                   */ \
    "v_mul_f32 %[W], %[X], -%[invd]\n\t"                      /*    W   = -U / d              (1/d written four slots back)       */ \
    "v_mul_f32 %[k], %[k], %[invd]\n\t"                       /*    k  /= d                                                       */ \
    "v_add_f32 %[pa], %[pA], %[Y]\n\t"                        /*    pa  = pA + IA c                                               */ \
    "v_fmac_f32 %[pa], %[k], %[X]\n\t"                        /*    pa += k U                                                     */ \
    "v_mov_b32_dpp %[sU], %[X] " MPPI_SW "\n\t"               /*    sU  = swap(U)                                                 */ \
    "v_fmac_f32 %[D0], %[W], %[X]\n\t"                        /*    D0 += W U                 Ia = IA - U U^T / d                 */ \
    "v_fmac_f32_dpp %[D1], %[X], %[W] " MPPI_R1 "\n\t"        /*    D1 += rot1(U) W                                               */ \
    "v_fmac_f32_dpp %[D2], %[X], %[W] " MPPI_R2 "\n\t"        /*    D2 += rot2(U) W                                               */
// root joint of a fixed-base tree: d, u, W = -U/d, k = u/d (nothing goes on to a parent)
#define MPPI_OCT_JOINT_ROOT                                                                                                          \
    "v_mul_f32 %[t1], %[S], %[X]\n\t"                         /*    t1  = S U                                                     */ \
    "v_mul_f32 %[t2], %[S], %[pA]\n\t"                        /*    t2  = S pA                                                    */ \
    "s_nop 0\n\t"                                                                                                                    \
    "v_add_f32_dpp %[t1], %[t1], %[t1] " MPPI_SW "\n\t"       /*    t1 += swap(t1)                                                */ \
    "v_add_f32_dpp %[t2], %[t2], %[t2] " MPPI_SW "\n\t"       /*    t2 += swap(t2)                                                */ \
    "s_nop 0\n\t"                                                                                                                    \
    "v_add_f32_dpp %[d], %[t1], %[kdh] " MPPI_B(0) "\n\t"                                                                            \
    "v_subrev_f32_dpp %[u], %[t2], %[tau] " MPPI_B(0) "\n\t"                                                                         \
    "v_add_f32_dpp %[d], %[t1], %[d] " MPPI_B(1) "\n\t"                                                                              \
    "v_subrev_f32_dpp %[u], %[t2], %[u] " MPPI_B(1) "\n\t"                                                                           \
    "v_add_f32_dpp %[d], %[t1], %[d] " MPPI_B(2) "\n\t"                                                                              \
    "v_subrev_f32_dpp %[u], %[t2], %[u] " MPPI_B(2) "\n\t"                                                                           \
    "v_rcp_f32 %[invd], %[d]\n\t"                                                                                                    \
    "s_nop 0\n\t"                                             /*    (a transcendental's result: not in the next slot)             */ \
    "v_mul_f32 %[W], %[X], -%[invd]\n\t"                      /*    W = -U / d                                                    */ \
    "v_mul_f32 %[k], %[u], %[invd]"                           /*    k = u / d                                                     */

// ---- pass 1: joint subspace, velocity, velocity-product bias of one body -------------------------------------------------------------
// S = (az | p x az): az in the angular lanes, rot1(p rot1(az) - rot1(p) az) in the linear ones (bank_mask 0xc: quads 2, 3 of a row);
// v = vp + qd S;  w = wp + qd az (angular velocity, replicated in both halves);
// cb = vp x (qd S):  angular lanes wp x sja,  linear lanes wp x sjl + vpl x sja   (Z = vp in the linear lanes, 0 in the angular)
__device__ __forceinline__ void opass1_fused(OF p, OF az, OF qd, OF vp, OF wp, OF lin, OF &S, OF &v, OF &w, OF &cb) {
    OF saz, t, Z, sj, tc;
    asm("v_mov_b32 %[S], %[az]\n\t"                               //  1 S   = az            (p, az: maybe written right in front of the block)
        "v_mul_f32 %[saz], %[qd], %[az]\n\t"                      //  2 saz = qd az
        "v_mul_f32_dpp %[t], %[az], %[p] " MPPI_R1 "\n\t"         //  3 t   = rot1(az) p
        "v_fmac_f32_dpp %[t], %[p], -%[az] " MPPI_R1 "\n\t"       //  4 t  -= rot1(p) az
        "v_mul_f32 %[Z], %[lin], %[vp]\n\t"                       //  5 Z   = lin vp
        "v_add_f32 %[w], %[wp], %[saz]\n\t"                       //  6 w   = wp + saz
        "v_mov_b32_dpp %[S], %[t] quad_perm:[1,2,0,1] row_mask:0xf bank_mask:0xc\n\t"   //  7 S (linear lanes) = rot1(t)     (t written at 4)
        "v_mul_f32 %[sj], %[qd], %[S]\n\t"                        //  8 sj  = qd S
        "v_add_f32 %[v], %[vp], %[sj]\n\t"                        //  9 v   = vp + sj
        "v_mul_f32_dpp %[tc], %[saz], %[Z] " MPPI_R1 "\n\t"       // 10 tc  = rot1(saz) Z    (saz written at 2)
        "v_fmac_f32_dpp %[tc], %[Z], -%[saz] " MPPI_R1 "\n\t"     // 11 tc -= rot1(Z) saz    (Z written at 5)
        "v_fmac_f32_dpp %[tc], %[sj], %[wp] " MPPI_R1 "\n\t"      // 12 tc += rot1(sj) wp    (sj written at 8)
        "v_fmac_f32_dpp %[tc], %[wp], -%[sj] " MPPI_R1 "\n\t"     // 13 tc -= rot1(wp) sj
        "s_nop 1\n\t"                                             // 14                      (tc written at 13)
        "v_mov_b32_dpp %[cb], %[tc] " MPPI_R1                     // 15 cb  = rot1(tc)
        : [S] "=&v"(S), [v] "=&v"(v), [w] "=&v"(w), [cb] "=&v"(cb), [saz] "=&v"(saz), [t] "=&v"(t), [Z] "=&v"(Z), [sj] "=&v"(sj), [tc] "=&v"(tc)
        : [p] "v"(p), [az] "v"(az), [qd] "v"(qd), [vp] "v"(vp), [wp] "v"(wp), [lin] "v"(lin));
}
// the root of a fixed-base tree: v = qd S, w = qd az, no bias
__device__ __forceinline__ void opass1_root_fused(OF p, OF az, OF qd, OF &S, OF &v, OF &w) {
    OF t;
    asm("v_mov_b32 %[S], %[az]\n\t"                               //  1 S   = az
        "v_mul_f32 %[w], %[qd], %[az]\n\t"                        //  2 w   = qd az
        "v_mul_f32_dpp %[t], %[az], %[p] " MPPI_R1 "\n\t"         //  3 t   = rot1(az) p
        "v_fmac_f32_dpp %[t], %[p], -%[az] " MPPI_R1 "\n\t"       //  4 t  -= rot1(p) az
        "s_nop 1\n\t"                                             //  5                      (t written at 4)
        "v_mov_b32_dpp %[S], %[t] quad_perm:[1,2,0,1] row_mask:0xf bank_mask:0xc\n\t"   //  6 S (linear lanes) = rot1(t)
        "v_mul_f32 %[v], %[qd], %[S]"                             //  7 v   = qd S
        : [S] "=&v"(S), [v] "=&v"(v), [w] "=&v"(w), [t] "=&v"(t)
        : [p] "v"(p), [az] "v"(az), [qd] "v"(qd));
}

// ---- inward pass of one body -----------------------------------------------------------------------------------------------------------
// in:  Tr = R Ic rows and h = m c from qmoments (both halves; the linear lanes' view of the inertia block has Ic = 0, 1/m = 0),
//      R0..R2 = columns of R, v / w spatial / angular velocity, S, cb, kdh, tau; INNER bodies: the children's articulated inertia
//      (aD rows as inputs, the cross rows preloaded in O) and bias force pacc
// out: W = -U/d, k = (u - U.c)/d for the outward pass; the articulated inertia (D, O) and bias force pa this body hands to its parent
//  1-13  rigid inertia rows D (I | m 1) about the world origin
// 14-..  g = D v + sg rot1(h x s), s = swap(v):  angular n = I w + h x vl, linear f = m vl + w x h;  cross rows O1 (+)= -sg rot2(h),
//        O2 (+)= sg rot1(h)  (skew(h) | -skew(h));  pA = rot1(w x g + A2 x gs), gs = swap(g):  (w x n + vl x f | w x f)
//        then (U, V) = IA (S, c), the joint (three 6-dots: product, + swap, three broadcast sums - bit-identical in all eight lanes),
//        and the rank-one update IA += W rot(U)
struct OctBodyIn {
    OF Tr0, Tr1, Tr2, h, R0, R1, R2, m, invm, v, w, S, cb, kdh, tau;
};
template <bool LEAF>
__device__ __forceinline__ void obody_fused(const OctBodyIn &b, const OctLane &ol, OF aD0, OF aD1, OF aD2, OF pacc, OAI &A, OF &W, OF &k, OF &pa) {
    OF cw, t, s, g, th, A2, gs, t3, pA, sx, sy, X, Y, t1, t2, tk, d, u, invd, sU;
    if constexpr (LEAF) {
        asm(MPPI_OCT_INERTIA MPPI_OCT_G
            "v_mul_f32_dpp %[O1], %[h], %[nsg] " MPPI_R2 "\n\t"       // 21 O1  = rot2(h) nsg
            "v_fmac_f32_dpp %[g], %[th], %[sg] " MPPI_R1 "\n\t"       // 22 g  += rot1(th) sg         (th written at 19)
            "v_mul_f32_dpp %[O2], %[h], %[sg] " MPPI_R1 "\n\t"        // 23 O2  = rot1(h) sg
            "v_mul_f32 %[X], %[D0], %[S]\n\t"                         // 24 X   = D0 S
            "v_mov_b32_dpp %[gs], %[g] " MPPI_SW "\n\t"               // 25 gs  = swap(g)             (g written at 22)
            MPPI_OCT_T3                                               // 26-29
            "v_mul_f32 %[Y], %[D0], %[cb]\n\t"                        // 30 Y   = D0 c
            "v_mov_b32_dpp %[sx], %[S] " MPPI_SW "\n\t"               // 31 sx  = swap(S)
            "v_mov_b32_dpp %[pA], %[t3] " MPPI_R1 "\n\t"              // 32 pA  = rot1(t3)            (t3 written at 29)
            "v_mov_b32_dpp %[sy], %[cb] " MPPI_SW "\n\t"              // 33 sy  = swap(c)
            "v_fmac_f32_dpp %[X], %[S], %[D1] " MPPI_R1 "\n\t"        // 34 X  += rot1(S) D1
            "v_fmac_f32_dpp %[Y], %[cb], %[D1] " MPPI_R1 "\n\t"       // 35 Y  += rot1(c) D1
            "v_fmac_f32_dpp %[X], %[S], %[D2] " MPPI_R2 "\n\t"        // 36 X  += rot2(S) D2
            "v_fmac_f32_dpp %[Y], %[cb], %[D2] " MPPI_R2 "\n\t"       // 37 Y  += rot2(c) D2
            "v_fmac_f32_dpp %[X], %[sx], %[O1] " MPPI_R1 "\n\t"       // 38 X  += rot1(sx) O1         (sx written at 31)
            "v_fmac_f32_dpp %[Y], %[sy], %[O1] " MPPI_R1 "\n\t"       // 39 Y  += rot1(sy) O1         (sy written at 33)
            "v_fmac_f32_dpp %[X], %[sx], %[O2] " MPPI_R2 "\n\t"       // 40 X  += rot2(sx) O2
            "v_fmac_f32_dpp %[Y], %[sy], %[O2] " MPPI_R2 "\n\t"       // 41 Y  += rot2(sy) O2
            MPPI_OCT_JOINT                                            // 42-65
            "v_mul_f32 %[O0], %[W], %[sU]\n\t"                        // 66 O0  = W sU                (sU written at 62)
            "v_fmac_f32_dpp %[O1], %[sU], %[W] " MPPI_R1 "\n\t"       // 67 O1 += rot1(sU) W
            "v_fmac_f32_dpp %[O2], %[sU], %[W] " MPPI_R2              // 68 O2 += rot2(sU) W
            : [D0] "=&v"(A.D[0]), [D1] "=&v"(A.D[1]), [D2] "=&v"(A.D[2]), [O0] "=&v"(A.O[0]), [O1] "=&v"(A.O[1]), [O2] "=&v"(A.O[2]), [W] "=&v"(W), [k] "=&v"(k),
              [pa] "=&v"(pa), [cw] "=&v"(cw), [t] "=&v"(t), [s] "=&v"(s), [g] "=&v"(g), [th] "=&v"(th), [A2] "=&v"(A2), [gs] "=&v"(gs), [t3] "=&v"(t3),
              [pA] "=&v"(pA), [sx] "=&v"(sx), [sy] "=&v"(sy), [X] "=&v"(X), [Y] "=&v"(Y), [t1] "=&v"(t1), [t2] "=&v"(t2), [tk] "=&v"(tk), [d] "=&v"(d),
              [u] "=&v"(u), [invd] "=&v"(invd), [sU] "=&v"(sU)
            : [Tr0] "v"(b.Tr0), [Tr1] "v"(b.Tr1), [Tr2] "v"(b.Tr2), [h] "v"(b.h), [R0] "v"(b.R0), [R1] "v"(b.R1), [R2] "v"(b.R2), [m] "v"(b.m), [invm] "v"(b.invm),
              [v] "v"(b.v), [w] "v"(b.w), [S] "v"(b.S), [cb] "v"(b.cb), [kdh] "v"(b.kdh), [tau] "v"(b.tau), [sg] "v"(ol.sg), [nsg] "v"(ol.nsg), [ang] "v"(ol.ang),
              [lin] "v"(ol.lin));
    } else {
        asm(MPPI_OCT_INERTIA MPPI_OCT_G
            "v_fmac_f32_dpp %[O1], %[h], %[nsg] " MPPI_R2 "\n\t"      // 21 O1 += rot2(h) nsg         (on top of the children's rows)
            "v_fmac_f32_dpp %[g], %[th], %[sg] " MPPI_R1 "\n\t"       // 22 g  += rot1(th) sg         (th written at 19)
            "v_fmac_f32_dpp %[O2], %[h], %[sg] " MPPI_R1 "\n\t"       // 23 O2 += rot1(h) sg
            "v_add_f32 %[D0], %[D0], %[aD0]\n\t"                      // 24 D0 += children's D0       (the rigid rows were last read at 17)
            "v_mov_b32_dpp %[gs], %[g] " MPPI_SW "\n\t"               // 25 gs  = swap(g)             (g written at 22)
            MPPI_OCT_T3                                               // 26-29
            "v_add_f32 %[D1], %[D1], %[aD1]\n\t"                      // 30 D1 += children's D1
            "v_add_f32 %[D2], %[D2], %[aD2]\n\t"                      // 31 D2 += children's D2
            "v_mov_b32_dpp %[pA], %[t3] " MPPI_R1 "\n\t"              // 32 pA  = rot1(t3)            (t3 written at 29)
            "v_mul_f32 %[X], %[D0], %[S]\n\t"                         // 33 X   = D0 S
            "v_mul_f32 %[Y], %[D0], %[cb]\n\t"                        // 34 Y   = D0 c
            "v_mov_b32_dpp %[sx], %[S] " MPPI_SW "\n\t"               // 35 sx  = swap(S)
            "v_mov_b32_dpp %[sy], %[cb] " MPPI_SW "\n\t"              // 36 sy  = swap(c)
            "v_add_f32 %[pA], %[pA], %[pacc]\n\t"                     // 37 pA += children's bias force
            "v_fmac_f32_dpp %[X], %[S], %[D1] " MPPI_R1 "\n\t"        // 38 X  += rot1(S) D1
            "v_fmac_f32_dpp %[Y], %[cb], %[D1] " MPPI_R1 "\n\t"       // 39 Y  += rot1(c) D1
            "v_fmac_f32_dpp %[X], %[S], %[D2] " MPPI_R2 "\n\t"        // 40 X  += rot2(S) D2
            "v_fmac_f32_dpp %[Y], %[cb], %[D2] " MPPI_R2 "\n\t"       // 41 Y  += rot2(c) D2
            "v_fmac_f32 %[X], %[O0], %[sx]\n\t"                       // 42 X  += O0 sx
            "v_fmac_f32 %[Y], %[O0], %[sy]\n\t"                       // 43 Y  += O0 sy
            "v_fmac_f32_dpp %[X], %[sx], %[O1] " MPPI_R1 "\n\t"       // 44 X  += rot1(sx) O1         (sx written at 35)
            "v_fmac_f32_dpp %[Y], %[sy], %[O1] " MPPI_R1 "\n\t"       // 45 Y  += rot1(sy) O1
            "v_fmac_f32_dpp %[X], %[sx], %[O2] " MPPI_R2 "\n\t"       // 46 X  += rot2(sx) O2
            "v_fmac_f32_dpp %[Y], %[sy], %[O2] " MPPI_R2 "\n\t"       // 47 Y  += rot2(sy) O2
            MPPI_OCT_JOINT                                            // 48-71
            "v_fmac_f32 %[O0], %[W], %[sU]\n\t"                       // 72 O0 += W sU                (sU written at 68)
            "v_fmac_f32_dpp %[O1], %[sU], %[W] " MPPI_R1 "\n\t"       // 73 O1 += rot1(sU) W
            "v_fmac_f32_dpp %[O2], %[sU], %[W] " MPPI_R2              // 74 O2 += rot2(sU) W
            : [D0] "=&v"(A.D[0]), [D1] "=&v"(A.D[1]), [D2] "=&v"(A.D[2]), [O0] "+v"(A.O[0]), [O1] "+v"(A.O[1]), [O2] "+v"(A.O[2]), [W] "=&v"(W), [k] "=&v"(k),
              [pa] "=&v"(pa), [cw] "=&v"(cw), [t] "=&v"(t), [s] "=&v"(s), [g] "=&v"(g), [th] "=&v"(th), [A2] "=&v"(A2), [gs] "=&v"(gs), [t3] "=&v"(t3),
              [pA] "=&v"(pA), [sx] "=&v"(sx), [sy] "=&v"(sy), [X] "=&v"(X), [Y] "=&v"(Y), [t1] "=&v"(t1), [t2] "=&v"(t2), [tk] "=&v"(tk), [d] "=&v"(d),
              [u] "=&v"(u), [invd] "=&v"(invd), [sU] "=&v"(sU)
            : [Tr0] "v"(b.Tr0), [Tr1] "v"(b.Tr1), [Tr2] "v"(b.Tr2), [h] "v"(b.h), [R0] "v"(b.R0), [R1] "v"(b.R1), [R2] "v"(b.R2), [m] "v"(b.m), [invm] "v"(b.invm),
              [v] "v"(b.v), [w] "v"(b.w), [S] "v"(b.S), [cb] "v"(b.cb), [kdh] "v"(b.kdh), [tau] "v"(b.tau), [sg] "v"(ol.sg), [nsg] "v"(ol.nsg), [ang] "v"(ol.ang),
              [lin] "v"(ol.lin), [aD0] "v"(aD0), [aD1] "v"(aD1), [aD2] "v"(aD2), [pacc] "v"(pacc));
    }
}
// ... the root of a fixed-base tree: only W and k leave the block
template <bool LEAF>
__device__ __forceinline__ void obody_root_fused(const OctBodyIn &b, const OctLane &ol, OF aD0, OF aD1, OF aD2, OF pacc, OF aO0, OF aO1, OF aO2, OF &W, OF &k) {
    OF D0, D1, D2, O1, O2, cw, t, s, g, th, A2, gs, t3, pA, sx, X, t1, t2, d, u, invd;
    if constexpr (LEAF) {
        asm(MPPI_OCT_INERTIA MPPI_OCT_G
            "v_mul_f32_dpp %[O1], %[h], %[nsg] " MPPI_R2 "\n\t"       // 21 O1  = rot2(h) nsg
            "v_fmac_f32_dpp %[g], %[th], %[sg] " MPPI_R1 "\n\t"       // 22 g  += rot1(th) sg
            "v_mul_f32_dpp %[O2], %[h], %[sg] " MPPI_R1 "\n\t"        // 23 O2  = rot1(h) sg
            "v_mul_f32 %[X], %[D0], %[S]\n\t"                         // 24 X   = D0 S
            "v_mov_b32_dpp %[gs], %[g] " MPPI_SW "\n\t"               // 25 gs  = swap(g)
            MPPI_OCT_T3                                               // 26-29
            "v_mov_b32_dpp %[sx], %[S] " MPPI_SW "\n\t"               // 30 sx  = swap(S)
            "v_fmac_f32_dpp %[X], %[S], %[D1] " MPPI_R1 "\n\t"        // 31 X  += rot1(S) D1
            "v_mov_b32_dpp %[pA], %[t3] " MPPI_R1 "\n\t"              // 32 pA  = rot1(t3)            (t3 written at 29)
            "v_fmac_f32_dpp %[X], %[S], %[D2] " MPPI_R2 "\n\t"        // 33 X  += rot2(S) D2
            "v_fmac_f32_dpp %[X], %[sx], %[O1] " MPPI_R1 "\n\t"       // 34 X  += rot1(sx) O1         (sx written at 30)
            "v_fmac_f32_dpp %[X], %[sx], %[O2] " MPPI_R2 "\n\t"       // 35 X  += rot2(sx) O2
            MPPI_OCT_JOINT_ROOT
            : [W] "=&v"(W), [k] "=&v"(k), [D0] "=&v"(D0), [D1] "=&v"(D1), [D2] "=&v"(D2), [O1] "=&v"(O1), [O2] "=&v"(O2), [cw] "=&v"(cw), [t] "=&v"(t), [s] "=&v"(s),
              [g] "=&v"(g), [th] "=&v"(th), [A2] "=&v"(A2), [gs] "=&v"(gs), [t3] "=&v"(t3), [pA] "=&v"(pA), [sx] "=&v"(sx), [X] "=&v"(X), [t1] "=&v"(t1),
              [t2] "=&v"(t2), [d] "=&v"(d), [u] "=&v"(u), [invd] "=&v"(invd)
            : [Tr0] "v"(b.Tr0), [Tr1] "v"(b.Tr1), [Tr2] "v"(b.Tr2), [h] "v"(b.h), [R0] "v"(b.R0), [R1] "v"(b.R1), [R2] "v"(b.R2), [m] "v"(b.m), [invm] "v"(b.invm),
              [v] "v"(b.v), [w] "v"(b.w), [S] "v"(b.S), [kdh] "v"(b.kdh), [tau] "v"(b.tau), [sg] "v"(ol.sg), [nsg] "v"(ol.nsg), [ang] "v"(ol.ang), [lin] "v"(ol.lin));
    } else {
        O1 = aO1;
        O2 = aO2;
        asm(MPPI_OCT_INERTIA MPPI_OCT_G
            "v_fmac_f32_dpp %[O1], %[h], %[nsg] " MPPI_R2 "\n\t"      // 21 O1 += rot2(h) nsg
            "v_fmac_f32_dpp %[g], %[th], %[sg] " MPPI_R1 "\n\t"       // 22 g  += rot1(th) sg
            "v_fmac_f32_dpp %[O2], %[h], %[sg] " MPPI_R1 "\n\t"       // 23 O2 += rot1(h) sg
            "v_add_f32 %[D0], %[D0], %[aD0]\n\t"                      // 24 D0 += children's D0
            "v_mov_b32_dpp %[gs], %[g] " MPPI_SW "\n\t"               // 25 gs  = swap(g)
            MPPI_OCT_T3                                               // 26-29
            "v_add_f32 %[D1], %[D1], %[aD1]\n\t"                      // 30 D1 += children's D1
            "v_add_f32 %[D2], %[D2], %[aD2]\n\t"                      // 31 D2 += children's D2
            "v_mov_b32_dpp %[pA], %[t3] " MPPI_R1 "\n\t"              // 32 pA  = rot1(t3)
            "v_mul_f32 %[X], %[D0], %[S]\n\t"                         // 33 X   = D0 S
            "v_mov_b32_dpp %[sx], %[S] " MPPI_SW "\n\t"               // 34 sx  = swap(S)
            "v_add_f32 %[pA], %[pA], %[pacc]\n\t"                     // 35 pA += children's bias force
            "v_fmac_f32_dpp %[X], %[S], %[D1] " MPPI_R1 "\n\t"        // 36 X  += rot1(S) D1
            "v_fmac_f32_dpp %[X], %[S], %[D2] " MPPI_R2 "\n\t"        // 37 X  += rot2(S) D2
            "v_fmac_f32 %[X], %[aO0], %[sx]\n\t"                      // 38 X  += O0 sx
            "v_fmac_f32_dpp %[X], %[sx], %[O1] " MPPI_R1 "\n\t"       // 39 X  += rot1(sx) O1         (sx written at 34)
            "v_fmac_f32_dpp %[X], %[sx], %[O2] " MPPI_R2 "\n\t"       // 40 X  += rot2(sx) O2
            MPPI_OCT_JOINT_ROOT
            : [W] "=&v"(W), [k] "=&v"(k), [D0] "=&v"(D0), [D1] "=&v"(D1), [D2] "=&v"(D2), [O1] "+v"(O1), [O2] "+v"(O2), [cw] "=&v"(cw), [t] "=&v"(t), [s] "=&v"(s),
              [g] "=&v"(g), [th] "=&v"(th), [A2] "=&v"(A2), [gs] "=&v"(gs), [t3] "=&v"(t3), [pA] "=&v"(pA), [sx] "=&v"(sx), [X] "=&v"(X), [t1] "=&v"(t1),
              [t2] "=&v"(t2), [d] "=&v"(d), [u] "=&v"(u), [invd] "=&v"(invd)
            : [Tr0] "v"(b.Tr0), [Tr1] "v"(b.Tr1), [Tr2] "v"(b.Tr2), [h] "v"(b.h), [R0] "v"(b.R0), [R1] "v"(b.R1), [R2] "v"(b.R2), [m] "v"(b.m), [invm] "v"(b.invm),
              [v] "v"(b.v), [w] "v"(b.w), [S] "v"(b.S), [kdh] "v"(b.kdh), [tau] "v"(b.tau), [sg] "v"(ol.sg), [nsg] "v"(ol.nsg), [ang] "v"(ol.ang), [lin] "v"(ol.lin),
              [aD0] "v"(aD0), [aD1] "v"(aD1), [aD2] "v"(aD2), [pacc] "v"(pacc), [aO0] "v"(aO0));
    }
}
// ---- outward pass of one joint: qdd = k + W . a_parent,  a = a_parent + c + qdd S --------------------------------------------------------
// (left to the compiler - the same nine operations in C++, DPP moves folded into the adds by its combiner - the substep loop grows
// from 1003 to 1012 instructions with 29 instead of 25 s_nop: it does not fill the wait slots with the integration's work either)
__device__ __forceinline__ void ooutward_fused(OF W, OF ap, OF cb, OF S, OF k, OF &dd, OF &a) {
    OF t;
    asm("v_mul_f32 %[t], %[W], %[ap]\n\t"                             //  1 t   = W ap
        "v_add_f32 %[a], %[ap], %[cb]\n\t"                            //  2 a   = ap + c
        "s_nop 0\n\t"                                                 //  3
        "v_add_f32_dpp %[t], %[t], %[t] " MPPI_SW "\n\t"              //  4 t  += swap(t)          (t written at 1)
        "s_nop 1\n\t"                                                 //  5
        "v_add_f32_dpp %[dd], %[t], %[k] " MPPI_B(0) "\n\t"           //  6 dd  = t[0] + k         (t written at 4)
        "v_add_f32_dpp %[dd], %[t], %[dd] " MPPI_B(1) "\n\t"          //  7 dd += t[1]
        "v_add_f32_dpp %[dd], %[t], %[dd] " MPPI_B(2) "\n\t"          //  8 dd += t[2]
        "v_fmac_f32 %[a], %[dd], %[S]"                                //  9 a  += dd S
        : [dd] "=&v"(dd), [a] "=&v"(a), [t] "=&v"(t)
        : [W] "v"(W), [ap] "v"(ap), [cb] "v"(cb), [S] "v"(S), [k] "v"(k));
}
// ... with the block's three wait slots FILLED with the drive-limit test of the PREVIOUS joint (quad_step: torque of the implicit
// drive tt = tau - kdh qdd, running maximum of |tt| - effort): three independent instructions that otherwise follow the pass as
// code of their own, here in the slots the DPP hazards leave empty - the same operations, the same results
__device__ __forceinline__ void ooutward_fused_check(OF W, OF ap, OF cb, OF S, OF k, OF &dd, OF &a, OF tau_p, OF kdh_p, OF dd_p, OF eff_p, OF &tt_p,
                                                     OF &excess) {
    OF t, e;
    asm("v_mul_f32 %[t], %[W], %[ap]\n\t"                             //  1 t   = W ap
        "v_add_f32 %[a], %[ap], %[cb]\n\t"                            //  2 a   = ap + c
        "v_fma_f32 %[ttp], -%[kdhp], %[ddp], %[taup]\n\t"             //  3 tt' = tau' - kdh' qdd'        (previous joint)
        "v_add_f32_dpp %[t], %[t], %[t] " MPPI_SW "\n\t"              //  4 t  += swap(t)          (t written at 1)
        "v_sub_f32 %[e], |%[ttp]|, %[effp]\n\t"                       //  5 e   = |tt'| - effort'
        "v_max_f32 %[ex], %[ex], %[e]\n\t"                            //  6 excess = max(excess, e)
        "v_add_f32_dpp %[dd], %[t], %[k] " MPPI_B(0) "\n\t"           //  7 dd  = t[0] + k         (t written at 4: two slots)
        "v_add_f32_dpp %[dd], %[t], %[dd] " MPPI_B(1) "\n\t"          //  8 dd += t[1]
        "v_add_f32_dpp %[dd], %[t], %[dd] " MPPI_B(2) "\n\t"          //  9 dd += t[2]
        "v_fmac_f32 %[a], %[dd], %[S]"                                // 10 a  += dd S
        : [dd] "=&v"(dd), [a] "=&v"(a), [t] "=&v"(t), [e] "=&v"(e), [ttp] "=&v"(tt_p), [ex] "+v"(excess)
        : [W] "v"(W), [ap] "v"(ap), [cb] "v"(cb), [S] "v"(S), [k] "v"(k), [taup] "v"(tau_p), [kdhp] "v"(kdh_p), [ddp] "v"(dd_p), [effp] "v"(eff_p));
}

// Articulated-body solve, octet-parallel: same interface and arithmetic as quad_aba (mppi_quad.hpp) up to the association of
// the sums.  `bodies`: this LANE's view of the model's body blocks - the angular lanes read the model's own, the linear lanes a
// copy whose inertia tensors and 1/m are zero (oct_lin_view), so that the rigid-inertia rows come out as I in one half and
// zero in the other without a select.  tau / kdh / qd / qdd: replicated scalars (same in all eight lanes of a sample).
// CHECK: the drive-limit test of quad_step rides in the wait slots of the outward pass (ooutward_fused_check): tt[i] = tau_exp[i] -
// kdh[i] qdd[i] and excess = max_i(|tt[i]| - effort_i) come back with the accelerations
template <class T, bool CHECK = false, class BP, class M, int JT>
__device__ __forceinline__ void oct_aba(M &m, BP bodies, const OctLane &ol, const QPose<T, JT> &P, const OF *qd, const OF *tau_exp, const OF *kdh, OF *qdd,
                                        JointLimits *lim, OF *tt = nullptr, OF *excess_out = nullptr) {
    constexpr int NB = T::NB;
    OF v[NB], w[NB], S[NB], cb[NB], W[NB], kk[NB], pacc[NB];
    OAI acc[NB];
    bool has_acc[NB];
    const OF zero = 0.f;
    // pass 1: subspaces, velocities, velocity-product biases, root to leaves
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        constexpr int pj = par < 0 ? 0 : par;
        const OF az = P.R2p[i].x;
        if (P.revolute(i)) {
            if constexpr (par < 0) {
                opass1_root_fused(P.pos(i), az, qd[i], S[i], v[i], w[i]);
                cb[i] = zero;
            } else {
                opass1_fused(P.pos(i), az, qd[i], v[pj], w[pj], ol.lin, S[i], v[i], w[i], cb[i]);
            }
        } else {  // prismatic: S = (0 | az), no angular velocity of its own; c = vp x (qd S) = (0 | wp x qd az)
            S[i] = ol.lin * az;
            const OF sj = qd[i] * S[i];
            if constexpr (par < 0) {
                v[i] = sj;
                w[i] = zero;
                cb[i] = zero;
            } else {
                v[i] = v[pj] + sj;
                w[i] = w[pj];
                cb[i] = qcross(w[pj], sj);
            }
        }
        has_acc[i] = false;
    });
    BodyK1 blk[NB];  // requested leaf-first, in the order the backward sweep consumes them
    static_rfor<0, NB>([&](auto ic) MPPI_LAMBDA { blk[ic] = load_block<BodyK1>(bodies[ic].k1); });
    static_rfor<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const BodyK1 &b = blk[i];
        lim[i] = {b.effort, b.vmax};
        OF h, Tr[3];
        qmoments(P.R01[i], P.R2p[i], b, h, Tr);   // h = R hb + m p (both halves), Tr = R Ic (zero in the linear lanes)
        const OctBodyIn in{Tr[0], Tr[1], Tr[2], h, P.R01[i].x, P.R01[i].y, P.R2p[i].x, b.m, b.invm, v[i], w[i], S[i], cb[i], kdh[i], tau_exp[i]};
        if constexpr (par < 0) {
            if (has_acc[i]) obody_root_fused<false>(in, ol, acc[i].D[0], acc[i].D[1], acc[i].D[2], pacc[i], acc[i].O[0], acc[i].O[1], acc[i].O[2], W[i], kk[i]);
            else obody_root_fused<true>(in, ol, zero, zero, zero, zero, zero, zero, zero, W[i], kk[i]);
        } else {
            OAI A;
            OF pa;
            if (has_acc[i]) {
                A.O[0] = acc[i].O[0]; A.O[1] = acc[i].O[1]; A.O[2] = acc[i].O[2];
                obody_fused<false>(in, ol, acc[i].D[0], acc[i].D[1], acc[i].D[2], pacc[i], A, W[i], kk[i], pa);
            } else {
                obody_fused<true>(in, ol, zero, zero, zero, zero, A, W[i], kk[i], pa);
            }
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
    // pass 3: accelerations, root to leaves.  Gravity = fictitious base acceleration -g (linear lanes)
    OF a[NB];
    OF a0 = zero;
    if (m.gravity_on) a0 = ol.lin * qsel(-m.g[0], -m.g[1], -m.g[2]);
    OF excess = -INFINITY;
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const OF apar = par >= 0 ? a[par < 0 ? 0 : par] : a0;
        if constexpr (CHECK && i > 0)
            ooutward_fused_check(W[i], apar, cb[i], S[i], kk[i], qdd[i], a[i], tau_exp[i - 1], kdh[i - 1], qdd[i - 1], OF(lim[i - 1].effort), tt[i - 1], excess);
        else
            ooutward_fused(W[i], apar, cb[i], S[i], kk[i], qdd[i], a[i]);
    });
    if constexpr (CHECK) {  // (the last joint's test has no block behind it)
        tt[NB - 1] = tau_exp[NB - 1] - kdh[NB - 1] * qdd[NB - 1];
        *excess_out = qmax(excess, qabs(tt[NB - 1]) - OF(lim[NB - 1].effort));
    }
}

// this lane's view of the body blocks (angular lanes: the model's own, linear lanes: the copy staged with oct_lin_view).  The
// pick is made opaque: left to the optimiser, "loads through a select of two constant addresses" become two specialised copies
// of every block read behind a divergent branch.
typedef const MPPI_LDS_AS DevBody *OctBodies;
__device__ __forceinline__ OctBodies oct_bodies(const MPPI_LDS_AS DevBody *model_bodies, const MPPI_LDS_AS DevBody *lin_bodies) {
    OctBodies p = oct_half() ? lin_bodies : model_bodies;
    asm volatile("" : "+v"(p));
    return p;
}
// Where the linear lanes' copy of the body blocks goes inside a raw LDS array with 256 bytes of slack: 128 bytes (mod 256) away
// from the model's own blocks.  A ds_read_b128 serves angular and linear lanes in the same lane group, i.e. TWO addresses per
// group; 64 banks x 4 B = 256 B, so copies at the same offset mod 256 would hit the same banks on every block read (measured
// with the copy placed by the compiler: 4.5 k SQ_LDS_BANK_CONFLICT cycles per wavefront where the quad layout has none).
constexpr int oct_lin_raw_bytes(int nb) { return nb * (int)sizeof(DevBody) + 256; }
__device__ __forceinline__ MPPI_LDS_AS DevBody *oct_lin_place(MPPI_LDS_AS void *raw, const MPPI_LDS_AS DevBody *model_bodies) {
    const unsigned r = (unsigned)(unsigned long)raw, b = (unsigned)(unsigned long)model_bodies;
    return (MPPI_LDS_AS DevBody *)(unsigned long)(r + ((b + 128u - r) & 255u));
}
// the solve policy of quad_step / quad_rollout (mppi_quad.hpp QuadAba) for the octet layout
struct OctAba {
    OctBodies bodies;
    OctLane ol;
    static constexpr bool kFusedLimitCheck = true;
    template <class T, class M, int JT>
    __device__ __forceinline__ void aba(M &m, const QPose<T, JT> &P, const QF *qd, const QF *tau_exp, const QF *kdh, QF *qdd, JointLimits *lim) const {
        oct_aba<T>(m, bodies, ol, P, qd, tau_exp, kdh, qdd, lim);
    }
    // ... and the drive-limit test of quad_step with it: tt[i] = tau_exp[i] - kdh[i] qdd[i], excess = max_i(|tt[i]| - effort_i)
    template <class T, class M, int JT>
    __device__ __forceinline__ void aba_checked(M &m, const QPose<T, JT> &P, const QF *qd, const QF *tau_exp, const QF *kdh, QF *qdd, JointLimits *lim,
                                                QF *tt, QF &excess) const {
        oct_aba<T, true>(m, bodies, ol, P, qd, tau_exp, kdh, qdd, lim, tt, &excess);
    }
};

// the linear lanes' view of a body's inertia block: inertia tensor and 1/m zero, everything else as it is
__device__ __forceinline__ BodyK1 oct_lin_view(const BodyK1 &b) {
    BodyK1 o = b;
    o.hI[1] = o.hI[3] = o.hI[5] = 0.f;
    for (int j = 0; j < 6; j++) o.II[j] = 0.f;
    o.invm = 0.f;
    return o;
}

}  // namespace mppi
#endif
