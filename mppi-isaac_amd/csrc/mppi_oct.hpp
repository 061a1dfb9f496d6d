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
// is left (s_nop); the result of a transcendental is not read in the next issue slot.
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

// ---- pass 1 ------------------------------------------------------------------------------------------------------------------
// (p x az) before its rotation: the linear part of a revolute joint's subspace is rot1 of this
__device__ __forceinline__ OF ocross_pre(OF p, OF az) {
    OF t;
    asm("v_mul_f32_dpp %0, %2, %1 " MPPI_R1 "\n\t"      // t  = rot1(az) p
        "v_fmac_f32_dpp %0, %1, -%2 " MPPI_R1             // t -= rot1(p) az
        : "=&v"(t)
        : "v"(p), "v"(az));
    return t;
}
// S = az in the angular lanes, rot1(t) in the linear ones (bank_mask 0xc: quads 2, 3 of every row are written)
template <bool WAIT>
__device__ __forceinline__ OF osubspace(OF az, OF t) {
    OF S = az;
    if constexpr (WAIT) asm volatile("s_nop 1");  // (a one-body tree: t was written right in front of this)
    asm("v_mov_b32_dpp %0, %1 quad_perm:[1,2,0,1] row_mask:0xf bank_mask:0xc" : "+v"(S) : "v"(t));
    return S;
}
// v = vp + qd S, w = wp + qd az (angular velocity, replicated in both halves) and the velocity-product bias
// cb = vp x (qd S):  angular lanes wp x sja, linear lanes wp x sjl + vpl x sja
__device__ __forceinline__ void ovel_bias_fused(OF vp, OF wp, OF S, OF az, OF qd, OF lin, OF &v, OF &w, OF &cb) {
    OF sj, saz, Z, tc;
    asm("v_mul_f32 %3, %11, %9\n\t"                      //  1 sj  = qd S
        "v_mul_f32 %4, %11, %10\n\t"                     //  2 saz = qd az
        "v_mul_f32 %5, %12, %7\n\t"                      //  3 Z   = lin vp           (parent's linear velocity, 0 in the angular lanes)
        "v_mul_f32_dpp %6, %3, %8 " MPPI_R1 "\n\t"       //  4 tc  = rot1(sj) wp      (sj written at 1)
        "v_fmac_f32_dpp %6, %8, -%3 " MPPI_R1 "\n\t"     //  5 tc -= rot1(wp) sj
        "v_fmac_f32_dpp %6, %4, %5 " MPPI_R1 "\n\t"      //  6 tc += rot1(saz) Z
        "v_fmac_f32_dpp %6, %5, -%4 " MPPI_R1 "\n\t"     //  7 tc -= rot1(Z) saz      (Z written at 3)
        "v_add_f32 %0, %7, %3\n\t"                       //  8 v   = vp + sj
        "v_add_f32 %1, %8, %4\n\t"                       //  9 w   = wp + saz
        "v_mov_b32_dpp %2, %6 " MPPI_R1                  // 10 cb  = rot1(tc)         (tc written at 7)
        : "=&v"(v), "=&v"(w), "=&v"(cb), "=&v"(sj), "=&v"(saz), "=&v"(Z), "=&v"(tc)
        : "v"(vp), "v"(wp), "v"(S), "v"(az), "v"(qd), "v"(lin));
}

// ---- inward pass -------------------------------------------------------------------------------------------------------------
// Velocity-product force of the RIGID body, pA = v x* (I v), and the cross blocks of its inertia.  D0..D2: own-half rows of the
// rigid inertia (I in the angular lanes, m 1 in the linear ones), h = m c (both halves), v / w: spatial / angular velocity.
//   g  = D v + sg rot1(h x s)        s = swap(v):  angular n = I w + h x vl,  linear f = m vl + w x h
//   pA = rot1( w x g + A2 x gs )     gs = swap(g), A2 = vl in the angular lanes, 0 in the linear:  (w x n + vl x f | w x f)
//   O1 (+)= -sg rot2(h),  O2 (+)= sg rot1(h)       skew(h) | -skew(h) in rotated rows, on top of the children's blocks (LEAF: none)
// The two wait states in front of the last rotation carry the first two additions of the children's own-half rows (LEAF: a wait).
template <bool LEAF>
__device__ __forceinline__ void obias_fused(OF &D0, OF &D1, OF D2, OF h, OF v, OF w, const OctLane &ol, OF aD0, OF aD1, OF &O1, OF &O2, OF &pA) {
    OF s, g, th, gs, A2, t3;
    if constexpr (LEAF) {
        asm("v_mov_b32_dpp %3, %13 " MPPI_SW "\n\t"          //  1 s   = swap(v)
            "v_mul_f32 %4, %9, %13\n\t"                      //  2 g   = D0 v
            "v_fmac_f32_dpp %4, %13, %10 " MPPI_R1 "\n\t"    //  3 g  += rot1(v) D1
            "v_mul_f32_dpp %5, %3, %12 " MPPI_R1 "\n\t"      //  4 th  = rot1(s) h          (s written at 1)
            "v_fmac_f32_dpp %5, %12, -%3 " MPPI_R1 "\n\t"    //  5 th -= rot1(h) s
            "v_fmac_f32_dpp %4, %13, %11 " MPPI_R2 "\n\t"    //  6 g  += rot2(v) D2
            "v_mul_f32 %7, %17, %3\n\t"                      //  7 A2  = ang s
            "v_fmac_f32_dpp %4, %5, %15 " MPPI_R1 "\n\t"     //  8 g  += rot1(th) sg        (th written at 5)
            "v_mul_f32_dpp %1, %12, %16 " MPPI_R2 "\n\t"     //  9 O1  = rot2(h) nsg
            "v_mul_f32_dpp %2, %12, %15 " MPPI_R1 "\n\t"     // 10 O2  = rot1(h) sg
            "v_mov_b32_dpp %6, %4 " MPPI_SW "\n\t"           // 11 gs  = swap(g)            (g written at 8)
            "v_mul_f32_dpp %8, %4, %14 " MPPI_R1 "\n\t"      // 12 t3  = rot1(g) w
            "v_fmac_f32_dpp %8, %14, -%4 " MPPI_R1 "\n\t"    // 13 t3 -= rot1(w) g
            "v_fmac_f32_dpp %8, %6, %7 " MPPI_R1 "\n\t"      // 14 t3 += rot1(gs) A2        (gs written at 11)
            "v_fmac_f32_dpp %8, %7, -%6 " MPPI_R1 "\n\t"     // 15 t3 -= rot1(A2) gs
            "s_nop 1\n\t"                                    // 16                          (t3 written at 15)
            "v_mov_b32_dpp %0, %8 " MPPI_R1                  // 17 pA  = rot1(t3)
            : "=&v"(pA), "=&v"(O1), "=&v"(O2), "=&v"(s), "=&v"(g), "=&v"(th), "=&v"(gs), "=&v"(A2), "=&v"(t3)
            : "v"(D0), "v"(D1), "v"(D2), "v"(h), "v"(v), "v"(w), "v"(ol.sg), "v"(ol.nsg), "v"(ol.ang));
    } else {
        asm("v_mov_b32_dpp %3, %13 " MPPI_SW "\n\t"          //  1 s   = swap(v)
            "v_mul_f32 %4, %9, %13\n\t"                      //  2 g   = D0 v
            "v_fmac_f32_dpp %4, %13, %10 " MPPI_R1 "\n\t"    //  3 g  += rot1(v) D1
            "v_mul_f32_dpp %5, %3, %12 " MPPI_R1 "\n\t"      //  4 th  = rot1(s) h
            "v_fmac_f32_dpp %5, %12, -%3 " MPPI_R1 "\n\t"    //  5 th -= rot1(h) s
            "v_fmac_f32_dpp %4, %13, %11 " MPPI_R2 "\n\t"    //  6 g  += rot2(v) D2
            "v_mul_f32 %7, %17, %3\n\t"                      //  7 A2  = ang s
            "v_fmac_f32_dpp %4, %5, %15 " MPPI_R1 "\n\t"     //  8 g  += rot1(th) sg
            "v_fmac_f32_dpp %1, %12, %16 " MPPI_R2 "\n\t"    //  9 O1 += rot2(h) nsg
            "v_fmac_f32_dpp %2, %12, %15 " MPPI_R1 "\n\t"    // 10 O2 += rot1(h) sg
            "v_mov_b32_dpp %6, %4 " MPPI_SW "\n\t"           // 11 gs  = swap(g)
            "v_mul_f32_dpp %8, %4, %14 " MPPI_R1 "\n\t"      // 12 t3  = rot1(g) w
            "v_fmac_f32_dpp %8, %14, -%4 " MPPI_R1 "\n\t"    // 13 t3 -= rot1(w) g
            "v_fmac_f32_dpp %8, %6, %7 " MPPI_R1 "\n\t"      // 14 t3 += rot1(gs) A2
            "v_fmac_f32_dpp %8, %7, -%6 " MPPI_R1 "\n\t"     // 15 t3 -= rot1(A2) gs
            "v_add_f32 %9, %9, %18\n\t"                      // 16 D0 += children's D0      (the rigid rows were last read at 6)
            "v_add_f32 %10, %10, %19\n\t"                    // 17 D1 += children's D1
            "v_mov_b32_dpp %0, %8 " MPPI_R1                  // 18 pA  = rot1(t3)           (t3 written at 15)
            : "=&v"(pA), "+v"(O1), "+v"(O2), "=&v"(s), "=&v"(g), "=&v"(th), "=&v"(gs), "=&v"(A2), "=&v"(t3), "+v"(D0), "+v"(D1)
            : "v"(D2), "v"(h), "v"(v), "v"(w), "v"(ol.sg), "v"(ol.nsg), "v"(ol.ang), "v"(aD0), "v"(aD1));
    }
}
// X = A x and Y = A y for one inertia (U = IA S and IA c of a joint), interleaved: the two products fill each other's waits
template <bool O0ZERO>
__device__ __forceinline__ void omul2_fused(const OAI &A, OF x, OF y, OF &X, OF &Y) {
    OF sx, sy;
    if constexpr (O0ZERO) {
        asm("v_mul_f32 %0, %4, %10\n\t"                      //  1 X  = D0 x
            "v_mul_f32 %1, %4, %11\n\t"                      //  2 Y  = D0 y
            "v_mov_b32_dpp %2, %10 " MPPI_SW "\n\t"          //  3 sx = swap(x)
            "v_mov_b32_dpp %3, %11 " MPPI_SW "\n\t"          //  4 sy = swap(y)
            "v_fmac_f32_dpp %0, %10, %5 " MPPI_R1 "\n\t"     //  5 X += rot1(x) D1
            "v_fmac_f32_dpp %1, %11, %5 " MPPI_R1 "\n\t"
            "v_fmac_f32_dpp %0, %10, %6 " MPPI_R2 "\n\t"     //  7 X += rot2(x) D2
            "v_fmac_f32_dpp %1, %11, %6 " MPPI_R2 "\n\t"
            "v_fmac_f32_dpp %0, %2, %8 " MPPI_R1 "\n\t"      //  9 X += rot1(sx) O1        (sx written at 3)
            "v_fmac_f32_dpp %1, %3, %8 " MPPI_R1 "\n\t"
            "v_fmac_f32_dpp %0, %2, %9 " MPPI_R2 "\n\t"      // 11 X += rot2(sx) O2
            "v_fmac_f32_dpp %1, %3, %9 " MPPI_R2
            : "=&v"(X), "=&v"(Y), "=&v"(sx), "=&v"(sy)
            : "v"(A.D[0]), "v"(A.D[1]), "v"(A.D[2]), "v"(A.O[0]), "v"(A.O[1]), "v"(A.O[2]), "v"(x), "v"(y));
    } else {
        asm("v_mul_f32 %0, %4, %10\n\t"                      //  1 X  = D0 x
            "v_mul_f32 %1, %4, %11\n\t"                      //  2 Y  = D0 y
            "v_mov_b32_dpp %2, %10 " MPPI_SW "\n\t"          //  3 sx = swap(x)
            "v_mov_b32_dpp %3, %11 " MPPI_SW "\n\t"          //  4 sy = swap(y)
            "v_fmac_f32_dpp %0, %10, %5 " MPPI_R1 "\n\t"     //  5 X += rot1(x) D1
            "v_fmac_f32_dpp %1, %11, %5 " MPPI_R1 "\n\t"
            "v_fmac_f32_dpp %0, %10, %6 " MPPI_R2 "\n\t"     //  7 X += rot2(x) D2
            "v_fmac_f32_dpp %1, %11, %6 " MPPI_R2 "\n\t"
            "v_fmac_f32 %0, %7, %2\n\t"                      //  9 X += O0 sx
            "v_fmac_f32 %1, %7, %3\n\t"
            "v_fmac_f32_dpp %0, %2, %8 " MPPI_R1 "\n\t"      // 11 X += rot1(sx) O1
            "v_fmac_f32_dpp %1, %3, %8 " MPPI_R1 "\n\t"
            "v_fmac_f32_dpp %0, %2, %9 " MPPI_R2 "\n\t"      // 13 X += rot2(sx) O2
            "v_fmac_f32_dpp %1, %3, %9 " MPPI_R2
            : "=&v"(X), "=&v"(Y), "=&v"(sx), "=&v"(sy)
            : "v"(A.D[0]), "v"(A.D[1]), "v"(A.D[2]), "v"(A.O[0]), "v"(A.O[1]), "v"(A.O[2]), "v"(x), "v"(y));
    }
}
// ... and a single product (the root of a fixed-base tree has nothing to hand on to a parent)
__device__ __forceinline__ OF omul_fused(const OAI &A, OF x) {
    OF X, sx;
    asm("v_mul_f32 %0, %2, %8\n\t"                           //  1 X  = D0 x
        "s_nop 0\n\t"                                        //  2                       (x may have been written just before the block)
        "v_mov_b32_dpp %1, %8 " MPPI_SW "\n\t"               //  3 sx = swap(x)
        "v_fmac_f32_dpp %0, %8, %3 " MPPI_R1 "\n\t"          //  4 X += rot1(x) D1
        "v_fmac_f32_dpp %0, %8, %4 " MPPI_R2 "\n\t"          //  5 X += rot2(x) D2
        "v_fmac_f32 %0, %5, %1\n\t"                          //  6 X += O0 sx
        "v_fmac_f32_dpp %0, %1, %6 " MPPI_R1 "\n\t"          //  7 X += rot1(sx) O1      (sx written at 3)
        "v_fmac_f32_dpp %0, %1, %7 " MPPI_R2                 //  8 X += rot2(sx) O2
        : "=&v"(X), "=&v"(sx)
        : "v"(A.D[0]), "v"(A.D[1]), "v"(A.D[2]), "v"(A.O[0]), "v"(A.O[1]), "v"(A.O[2]), "v"(x));
    return X;
}
// One joint of the inward pass after U = IA S, V = IA c:  d = kdh + S.U,  u = tau - S.pA,  W = -U/d,  k = (u - U.c)/d,
// pa = pA + V + k U.  Three 6-dots: product, + swap (the other half's three terms), then the three components from broadcasts -
// in every lane alike, so the replicated scalars are bit-identical over the eight lanes; the dots fill each other's waits.
__device__ __forceinline__ void ojoint_parent_fused(OF S, OF U, OF pA, OF cb, OF V, OF kdh, OF tau, OF &W, OF &k, OF &pa) {
    OF t1, t2, tk, d, u, invd;
    asm("v_mul_f32 %3, %9, %10\n\t"                          //  1 t1  = S U
        "v_mul_f32 %4, %9, %11\n\t"                          //  2 t2  = S pA
        "v_mul_f32 %5, %10, %12\n\t"                         //  3 tk  = U c
        "v_add_f32_dpp %3, %3, %3 " MPPI_SW "\n\t"           //  4 t1 += swap(t1)        (t1 written at 1)
        "v_add_f32_dpp %4, %4, %4 " MPPI_SW "\n\t"           //  5 t2 += swap(t2)
        "v_add_f32_dpp %5, %5, %5 " MPPI_SW "\n\t"           //  6 tk += swap(tk)
        "v_add_f32_dpp %6, %3, %14 " MPPI_B(0) "\n\t"        //  7 d   = t1[0] + kdh     (t1 written at 4)
        "v_subrev_f32_dpp %7, %4, %15 " MPPI_B(0) "\n\t"     //  8 u   = tau - t2[0]     (t2 written at 5)
        "v_add_f32_dpp %6, %3, %6 " MPPI_B(1) "\n\t"         //  9 d  += t1[1]
        "v_subrev_f32_dpp %7, %4, %7 " MPPI_B(1) "\n\t"      // 10 u  -= t2[1]
        "v_add_f32_dpp %6, %3, %6 " MPPI_B(2) "\n\t"         // 11 d  += t1[2]
        "v_subrev_f32_dpp %7, %4, %7 " MPPI_B(2) "\n\t"      // 12 u  -= t2[2]
        "v_rcp_f32 %8, %6\n\t"                               // 13 1/d
        "v_subrev_f32_dpp %1, %5, %7 " MPPI_B(0) "\n\t"      // 14 k   = u - tk[0]
        "v_subrev_f32_dpp %1, %5, %1 " MPPI_B(1) "\n\t"      // 15 k  -= tk[1]
        "v_subrev_f32_dpp %1, %5, %1 " MPPI_B(2) "\n\t"      // 16 k  -= tk[2]
        "v_mul_f32 %0, %10, -%8\n\t"                         // 17 W   = -U / d          (1/d written at 13)
        "v_mul_f32 %1, %1, %8\n\t"                           // 18 k  /= d
        "v_add_f32 %2, %11, %13\n\t"                         // 19 pa  = pA + V
        "v_fmac_f32 %2, %1, %10"                             // 20 pa += k U
        : "=&v"(W), "=&v"(k), "=&v"(pa), "=&v"(t1), "=&v"(t2), "=&v"(tk), "=&v"(d), "=&v"(u), "=&v"(invd)
        : "v"(S), "v"(U), "v"(pA), "v"(cb), "v"(V), "v"(kdh), "v"(tau));
}
// the root joint of a fixed-base tree: d, u, W and k = u / d (its velocity-product bias is zero)
__device__ __forceinline__ void ojoint_root_fused(OF S, OF U, OF pA, OF kdh, OF tau, OF &W, OF &k) {
    OF t1, t2, d, u, invd;
    asm("v_mul_f32 %2, %7, %8\n\t"                           //  1 t1  = S U
        "v_mul_f32 %3, %7, %9\n\t"                           //  2 t2  = S pA
        "s_nop 0\n\t"                                        //  3
        "v_add_f32_dpp %2, %2, %2 " MPPI_SW "\n\t"           //  4 t1 += swap(t1)
        "v_add_f32_dpp %3, %3, %3 " MPPI_SW "\n\t"           //  5 t2 += swap(t2)
        "s_nop 0\n\t"                                        //  6
        "v_add_f32_dpp %4, %2, %10 " MPPI_B(0) "\n\t"        //  7 d   = t1[0] + kdh
        "v_subrev_f32_dpp %5, %3, %11 " MPPI_B(0) "\n\t"     //  8 u   = tau - t2[0]
        "v_add_f32_dpp %4, %2, %4 " MPPI_B(1) "\n\t"
        "v_subrev_f32_dpp %5, %3, %5 " MPPI_B(1) "\n\t"
        "v_add_f32_dpp %4, %2, %4 " MPPI_B(2) "\n\t"
        "v_subrev_f32_dpp %5, %3, %5 " MPPI_B(2) "\n\t"      // 12
        "v_rcp_f32 %6, %4\n\t"                               // 13 1/d
        "s_nop 0\n\t"                                        // 14                        (a transcendental's result: not in the next slot)
        "v_mul_f32 %0, %8, -%6\n\t"                          // 15 W = -U / d
        "v_mul_f32 %1, %5, %6"                               // 16 k = u / d
        : "=&v"(W), "=&v"(k), "=&v"(t1), "=&v"(t2), "=&v"(d), "=&v"(u), "=&v"(invd)
        : "v"(S), "v"(U), "v"(pA), "v"(kdh), "v"(tau));
}
// IA += W rot(U):  D[j] += W rot_j(U),  O[j] += W rot_j(swap U)      (Ia = IA - U U^T / d with W = -U / d)
template <bool O0ZERO>
__device__ __forceinline__ void orank1_fused(OAI &A, OF W, OF U) {
    OF sU;
    if constexpr (O0ZERO) {
        asm("v_mov_b32_dpp %6, %8 " MPPI_SW "\n\t"           //  1 sU  = swap(U)
            "v_fmac_f32 %0, %7, %8\n\t"                      //  2 D0 += W U
            "v_fmac_f32_dpp %1, %8, %7 " MPPI_R1 "\n\t"      //  3 D1 += rot1(U) W
            "v_fmac_f32_dpp %2, %8, %7 " MPPI_R2 "\n\t"      //  4 D2 += rot2(U) W
            "v_mul_f32 %3, %7, %6\n\t"                       //  5 O0  = W sU
            "v_fmac_f32_dpp %4, %6, %7 " MPPI_R1 "\n\t"      //  6 O1 += rot1(sU) W       (sU written at 1)
            "v_fmac_f32_dpp %5, %6, %7 " MPPI_R2             //  7 O2 += rot2(sU) W
            : "+v"(A.D[0]), "+v"(A.D[1]), "+v"(A.D[2]), "=&v"(A.O[0]), "+v"(A.O[1]), "+v"(A.O[2]), "=&v"(sU)
            : "v"(W), "v"(U));
    } else {
        asm("v_mov_b32_dpp %6, %8 " MPPI_SW "\n\t"
            "v_fmac_f32 %0, %7, %8\n\t"
            "v_fmac_f32_dpp %1, %8, %7 " MPPI_R1 "\n\t"
            "v_fmac_f32_dpp %2, %8, %7 " MPPI_R2 "\n\t"
            "v_fmac_f32 %3, %7, %6\n\t"                      //  5 O0 += W sU
            "v_fmac_f32_dpp %4, %6, %7 " MPPI_R1 "\n\t"
            "v_fmac_f32_dpp %5, %6, %7 " MPPI_R2
            : "+v"(A.D[0]), "+v"(A.D[1]), "+v"(A.D[2]), "+v"(A.O[0]), "+v"(A.O[1]), "+v"(A.O[2]), "=&v"(sU)
            : "v"(W), "v"(U));
    }
}
// outward pass of one joint: qdd = k + W . a_parent,  a = a_parent + c + qdd S
__device__ __forceinline__ void ooutward_fused(OF W, OF ap, OF cb, OF S, OF k, OF &dd, OF &a) {
    OF t;
    asm("v_mul_f32 %2, %3, %4\n\t"                           //  1 t   = W ap
        "v_add_f32 %1, %4, %5\n\t"                           //  2 a   = ap + c
        "s_nop 0\n\t"                                        //  3
        "v_add_f32_dpp %2, %2, %2 " MPPI_SW "\n\t"           //  4 t  += swap(t)          (t written at 1)
        "s_nop 1\n\t"                                        //  5
        "v_add_f32_dpp %0, %2, %7 " MPPI_B(0) "\n\t"         //  6 dd  = t[0] + k         (t written at 4)
        "v_add_f32_dpp %0, %2, %0 " MPPI_B(1) "\n\t"         //  7 dd += t[1]
        "v_add_f32_dpp %0, %2, %0 " MPPI_B(2) "\n\t"         //  8 dd += t[2]
        "v_fmac_f32 %1, %0, %6"                              //  9 a  += dd S
        : "=&v"(dd), "=&v"(a), "=&v"(t)
        : "v"(W), "v"(ap), "v"(cb), "v"(S), "v"(k));
}

// Articulated-body solve, octet-parallel: same interface and arithmetic as quad_aba (mppi_quad.hpp) up to the association of
// the sums.  `bodies`: this LANE's view of the model's body blocks - the angular lanes read the model's own, the linear lanes a
// copy whose inertia tensors and 1/m are zero (oct_stage_lin_view), so that the rigid-inertia block yields I in one half and
// nothing in the other without a select.  tau / kdh / qd / qdd: replicated scalars (same in all eight lanes of a sample).
template <class T, class BP, class M, int JT>
__device__ __forceinline__ void oct_aba(M &m, BP bodies, const OctLane &ol, const QPose<T, JT> &P, const OF *qd, const OF *tau_exp, const OF *kdh, OF *qdd,
                                        JointLimits *lim) {
    constexpr int NB = T::NB;
    OF v[NB], w[NB], S[NB], cb[NB], W[NB], kk[NB], pacc[NB];
    OAI acc[NB];
    bool has_acc[NB];
    const OF zero = 0.f;
    // pass 1: subspaces (products of all joints first: their common rotation reads its operand through DPP), velocities, biases
    OF St[NB];
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        St[i] = P.revolute(i) ? ocross_pre(P.pos(i), P.R2p[i].x) : zero;
    });
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        const OF az = P.R2p[i].x;
        // revolute: (az | p x az);  prismatic: (0 | az)
        S[i] = P.revolute(i) ? osubspace<(NB < 2)>(az, St[i]) : ol.lin * az;
    });
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const OF az = P.revolute(i) ? P.R2p[i].x : zero;
        if constexpr (par < 0) {
            v[i] = qd[i] * S[i];
            w[i] = qd[i] * az;
            cb[i] = zero;
        } else {
            constexpr int pj = par < 0 ? 0 : par;
            ovel_bias_fused(v[pj], w[pj], S[i], az, qd[i], ol.lin, v[i], w[i], cb[i]);
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
        const QM3 R = P.rot(i);
        // rigid inertia about the world origin (angular lanes; the linear view's Ic = 0, 1/m = 0 leave zero there) + m 1 (linear)
        OF h, Tr[3];
        qmoments(P.R01[i], P.R2p[i], b, h, Tr);
        const OF cw = b.invm * h;
        OAI A;
        qinertia_rows_fused(Tr, R.c, h, cw, A.D[0], A.D[1], A.D[2]);
        A.D[0] += ol.lin * b.m;
        OF pA;
        if (has_acc[i]) {
            A.O[0] = acc[i].O[0]; A.O[1] = acc[i].O[1]; A.O[2] = acc[i].O[2];
            obias_fused<false>(A.D[0], A.D[1], A.D[2], h, v[i], w[i], ol, acc[i].D[0], acc[i].D[1], A.O[1], A.O[2], pA);
            A.D[2] += acc[i].D[2];
            pA += pacc[i];
        } else {
            A.O[0] = zero;
            obias_fused<true>(A.D[0], A.D[1], A.D[2], h, v[i], w[i], ol, zero, zero, A.O[1], A.O[2], pA);
        }
        if constexpr (par < 0) {
            const OF Ui = omul_fused(A, S[i]);
            ojoint_root_fused(S[i], Ui, pA, kdh[i], tau_exp[i], W[i], kk[i]);
        } else {
            OF Ui, Vi, pa;
            if (has_acc[i]) omul2_fused<false>(A, S[i], cb[i], Ui, Vi);
            else omul2_fused<true>(A, S[i], cb[i], Ui, Vi);
            ojoint_parent_fused(S[i], Ui, pA, cb[i], Vi, kdh[i], tau_exp[i], W[i], kk[i], pa);
            if (has_acc[i]) orank1_fused<false>(A, W[i], Ui);
            else orank1_fused<true>(A, W[i], Ui);
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
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const OF apar = par >= 0 ? a[par < 0 ? 0 : par] : a0;
        ooutward_fused(W[i], apar, cb[i], S[i], kk[i], qdd[i], a[i]);
    });
}

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
