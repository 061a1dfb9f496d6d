// mppi_quad.hpp - quad-parallel rollout: ONE SAMPLE PER 4-LANE QUAD (fixed-base, contact-free scenes).
//
// Why: at K = 4096 the one-lane-per-sample kernel is 64 wavefronts on a 1024-SIMD chip, and a lone
// wavefront issues one instruction per ~4 cycles whatever its ILP (measured: SQ_ACTIVE_INST_ANY =
// 73 % of wave cycles at 1.04 instructions per quad-cycle) - the kernel is bound by ITS OWN
// instruction count.  Splitting a sample over the x/y/z components of its 3-vectors cuts the per-lane
// instruction stream ~3x and quadruples the wavefronts in flight.
//
// Layout inside a quad: lanes 0,1,2 own component (matrix row) r = 0,1,2; lane 3 is an exact replica of
// lane 0 (same data, same permutation sources) so wave-uniform branches and replicated scalars never
// diverge and nothing undefined is ever read.  Cross-lane traffic is DPP quad_perm only (a VALU operand
// modifier - no LDS, no extra latency):
//   rot1/rot2   lane r reads component (r+1)%3 / (r+2)%3        cross products, rotated-row matvecs
//   bc<k>       every lane reads component k                    standard-row products
// 3x3 pose matrices are stored as STANDARD rows (row r = R[r][0..2]) because they multiply uniform
// constant matrices; the four 3x3 blocks of an articulated inertia are stored as ROTATED rows
// (a[j] = A[r][(r+j)%3]) because they only ever meet distributed vectors, and skew(h) is natural there.
// H^T is kept alongside H (column access is the one thing a row distribution cannot do cheaply).
//
// The same templates compile for the host (tests/hostemu): QF is then a 4-float struct and the
// permutations are array shuffles, so the arithmetic is checked against the oracle without a GPU.
#pragma once
#include <cstddef>

#include "mppi_device.hpp"

namespace mppi {

#if defined(__HIP_DEVICE_COMPILE__)
typedef float QF;
template <int CTRL>
__device__ __forceinline__ float quad_dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ QF rot1(QF x) { return quad_dpp<0x49>(x); }  // quad_perm [1,2,0,1]
__device__ __forceinline__ QF rot2(QF x) { return quad_dpp<0x92>(x); }  // quad_perm [2,0,1,2]
template <int K>
__device__ __forceinline__ QF bc(QF x) { return quad_dpp<K * 0x55>(x); }  // quad_perm [k,k,k,k]
__device__ __forceinline__ int quad_row() { int l = threadIdx.x & 3; return l == 3 ? 0 : l; }
__device__ __forceinline__ QF qsel(float x0, float x1, float x2) { int r = quad_row(); return r == 0 ? x0 : (r == 1 ? x1 : x2); }
__device__ __forceinline__ QF qrep(float x) { return x; }
__device__ __forceinline__ float qlane0(QF x) { return x; }  // value as seen by the calling lane
__device__ __forceinline__ QF qsqrt(QF x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32 (1 ulp) instead of the 22-instruction IEEE sqrtf
// reciprocal: v_rcp_f32 (1 ulp) + one Newton step (3 instructions instead of the ~10 of an IEEE division)
__device__ __forceinline__ QF qrcp(QF x) { float r = __builtin_amdgcn_rcpf(x); return r * (2.f - x * r); }
__device__ __forceinline__ QF qabs(QF x) { return fabsf(x); }
__device__ __forceinline__ QF qclamp(QF v, QF lo, QF hi) { return __builtin_amdgcn_fmed3f(v, lo, hi); }
__device__ __forceinline__ QF qmin(QF a, QF b) { return fminf(a, b); }
__device__ __forceinline__ QF qmax(QF a, QF b) { return fmaxf(a, b); }
#if defined(MPPI_FAST_ATAN)   // experiment build (variant table of the headline kernel)
__device__ __forceinline__ QF qatan2(QF a, QF b) { return fast_atan2(a, b); }
__device__ __forceinline__ QF qasin(QF a) { return fast_asin(a); }
#else
__device__ __forceinline__ QF qatan2(QF a, QF b) { return atan2f(a, b); }
__device__ __forceinline__ QF qasin(QF a) { return asinf(a); }
#endif
// hardware v_sin_f32 / v_cos_f32: measured max abs error 6.6e-7 / 4.4e-7 on [-6.4, 6.4] (tools/exp/sin_acc.hip),
// 3 instructions instead of ~28 for the polynomial version the host build keeps
__device__ __forceinline__ void qsincos(QF x, QF &s, QF &c) { s = __sinf(x); c = __cosf(x); }
__device__ __forceinline__ bool qany_gt(QF a, QF b) { return a > b; }     // replicated scalars: same in every lane of the quad
__device__ __forceinline__ QF qwhere_gt(QF a, QF b, QF x, QF y) { return a > b ? x : y; }
__device__ __forceinline__ QF qwhere_lt(QF a, QF b, QF x, QF y) { return a < b ? x : y; }
#else
struct QF {
    float v[4];
};
inline QF operator+(QF a, QF b) { QF o; for (int i = 0; i < 4; i++) o.v[i] = a.v[i] + b.v[i]; return o; }
inline QF operator-(QF a, QF b) { QF o; for (int i = 0; i < 4; i++) o.v[i] = a.v[i] - b.v[i]; return o; }
inline QF operator*(QF a, QF b) { QF o; for (int i = 0; i < 4; i++) o.v[i] = a.v[i] * b.v[i]; return o; }
inline QF operator*(float a, QF b) { QF o; for (int i = 0; i < 4; i++) o.v[i] = a * b.v[i]; return o; }
inline QF operator*(QF b, float a) { return a * b; }
inline QF operator+(QF b, float a) { QF o; for (int i = 0; i < 4; i++) o.v[i] = b.v[i] + a; return o; }
inline QF operator+(float a, QF b) { return b + a; }
inline QF operator-(QF b, float a) { return b + (-a); }
inline QF operator-(float a, QF b) { QF o; for (int i = 0; i < 4; i++) o.v[i] = a - b.v[i]; return o; }
inline QF operator-(QF a) { QF o; for (int i = 0; i < 4; i++) o.v[i] = -a.v[i]; return o; }
inline QF &operator+=(QF &a, QF b) { a = a + b; return a; }
inline QF &operator-=(QF &a, QF b) { a = a - b; return a; }
inline QF rot1(QF x) { return QF{{x.v[1], x.v[2], x.v[0], x.v[1]}}; }
inline QF rot2(QF x) { return QF{{x.v[2], x.v[0], x.v[1], x.v[2]}}; }
template <int K>
inline QF bc(QF x) { return QF{{x.v[K], x.v[K], x.v[K], x.v[K]}}; }
inline int quad_row() { return 0; }
inline QF qsel(float x0, float x1, float x2) { return QF{{x0, x1, x2, x0}}; }
inline QF qrep(float x) { return QF{{x, x, x, x}}; }
inline float qlane0(QF x) { return x.v[0]; }
inline QF qsqrt(QF x) { QF o; for (int i = 0; i < 4; i++) o.v[i] = sqrtf(x.v[i]); return o; }
inline QF qrcp(QF x) { QF o; for (int i = 0; i < 4; i++) o.v[i] = 1.f / x.v[i]; return o; }
inline QF qabs(QF x) { QF o; for (int i = 0; i < 4; i++) o.v[i] = fabsf(x.v[i]); return o; }
inline QF qclamp(QF v, QF lo, QF hi) { QF o; for (int i = 0; i < 4; i++) o.v[i] = fminf(fmaxf(v.v[i], lo.v[i]), hi.v[i]); return o; }
inline QF qmin(QF a, QF b) { QF o; for (int i = 0; i < 4; i++) o.v[i] = fminf(a.v[i], b.v[i]); return o; }
inline QF qmax(QF a, QF b) { QF o; for (int i = 0; i < 4; i++) o.v[i] = fmaxf(a.v[i], b.v[i]); return o; }
inline QF qatan2(QF a, QF b) { QF o; for (int i = 0; i < 4; i++) o.v[i] = atan2f(a.v[i], b.v[i]); return o; }
inline QF qasin(QF a) { QF o; for (int i = 0; i < 4; i++) o.v[i] = asinf(a.v[i]); return o; }
inline void qsincos(QF x, QF &s, QF &c) { for (int i = 0; i < 4; i++) fast_sincos(x.v[i], s.v[i], c.v[i]); }
inline bool qany_gt(QF a, QF b) { return a.v[0] > b.v[0]; }
inline QF qwhere_gt(QF a, QF b, QF x, QF y) { QF o; for (int i = 0; i < 4; i++) o.v[i] = a.v[i] > b.v[i] ? x.v[i] : y.v[i]; return o; }
inline QF qwhere_lt(QF a, QF b, QF x, QF y) { QF o; for (int i = 0; i < 4; i++) o.v[i] = a.v[i] < b.v[i] ? x.v[i] : y.v[i]; return o; }
#endif

// sum over the three components, taken from lane 0 so that replicated scalars (d, u, qdd, costs) are
// bit-identical across the quad whatever the summation order of each lane would have been
// (as (x0 + x1) + x2 from three broadcasts: the same value as bc<0>(x + rot1(x) + rot2(x)), and only the first of the three
// DPP reads can follow the write of x - one hazard wait instead of two on the dependent chains the dots sit in)
MPPI_HD QF qsum(QF x) { return bc<0>(x + rot1(x) + rot2(x)); }
// (a x b)_r = t_{r+1} with t_r = a_r b_{r+1} - a_{r+1} b_r: three permutations instead of four, two of
// them foldable into the multiply as DPP source modifiers
MPPI_HD QF qcross(QF a, QF b) { return rot1(a * rot1(b) - rot1(a) * b); }

struct QSV {  // spatial vector (world axes, about the world origin), one component per lane
    QF a, l;
};
struct QM3 {  // 3x3, STANDARD row r per lane
    QF c[3];
};
struct QAI {  // symmetric 6x6 [[I,H],[H^T,M]], ROTATED rows: x[j] = X[r][(r+j)%3]
    QF I[3], H[3], Ht[3], M[3];
};
// Two distributed values in ONE aligned register pair: the operand shape of the packed fp32 instructions (v_pk_mul_f32,
// v_pk_fma_f32: two multiply-adds per issue slot, each source half selectable by op_sel - a lone wavefront pays one issue slot
// of ~4.75 cycles per instruction whatever it does, tools/exp/issue_rate.hip).
#if defined(__HIP_DEVICE_COMPILE__)
typedef float QF2 __attribute__((ext_vector_type(2)));
#else
struct QF2 {
    QF x, y;
};
#endif


// ---- rotations folded into the multiply-adds (device only) -----------------------------------------------------------------
// The quad layout's matrix products multiply ROTATED copies of a distributed vector.  The compiler folds a quad_perm into a
// multiply or an add (v_mul_f32_dpp, v_add_f32_dpp) but never into a fused multiply-add: at the time its DPP combiner runs the
// operation is still the three-address v_fma_f32, which has no DPP form on gfx9; it becomes the two-address v_fmac_f32 (which
// has one) only after register allocation.  So every rotated operand of an FMA chain cost a v_mov_b32_dpp of its own - 450 of
// the 3 414 vector instructions of one horizon step of k_rollout_quad (ISA listing), on a kernel that is bound by exactly that
// count.  These blocks issue v_fmac_f32_dpp themselves.  Hazard: a DPP operand must not have been written by the two preceding
// VALU instructions (the compiler pads its own with s_nop and does not look into inline assembly): every block starts with at
// least two plain instructions and only ever reads its INPUTS through DPP, so no padding is needed.
// Same arithmetic as the C++ forms below them up to the association of the sums (last-bit differences).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MPPI_NO_DPP_FMAC)
#define MPPI_DPP_FMAC 1
// A block whose FIRST instructions are DPP reads is only safe where the compiler does not write one of their operands right
// in front of it (a copy, a reload from the AGPR file): the register-starved contact-scene kernels do, so their translation
// units define MPPI_DPP_LEAD_WAIT and such blocks start with the two wait states; the contact-free kernels go without, and
// tests/test_dpp_hazards.py checks every DPP instruction of the built library either way.
#if defined(MPPI_DPP_LEAD_WAIT)
#define MPPI_LEAD "s_nop 1\n\t"
#else
#define MPPI_LEAD ""
#endif
#define MPPI_R1 "quad_perm:[1,2,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define MPPI_R2 "quad_perm:[2,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#if defined(MPPI_NO_PK_QMUL)  // (A/B switch: the scalar forms)
__device__ __forceinline__ QSV qmul_fused(const QF *I, const QF *H, const QF *Ht, const QF *M, QF xa, QF xl) {
    QF ya, yl;
    asm("v_mul_f32 %0, %2, %14\n\t"                 // ya  = I0 xa
        "v_mul_f32 %1, %8, %14\n\t"                 // yl  = Ht0 xa
        "v_fmac_f32_dpp %0, %14, %3 " MPPI_R1 "\n\t"   // ya += rot1(xa) I1
        "v_fmac_f32_dpp %1, %14, %9 " MPPI_R1 "\n\t"   // yl += rot1(xa) Ht1
        "v_fmac_f32_dpp %0, %14, %4 " MPPI_R2 "\n\t"   // ya += rot2(xa) I2
        "v_fmac_f32_dpp %1, %14, %10 " MPPI_R2 "\n\t"  // yl += rot2(xa) Ht2
        "v_fmac_f32 %0, %5, %15\n\t"                // ya += H0 xl
        "v_fmac_f32 %1, %11, %15\n\t"               // yl += M0 xl
        "v_fmac_f32_dpp %0, %15, %6 " MPPI_R1 "\n\t"   // ya += rot1(xl) H1
        "v_fmac_f32_dpp %1, %15, %12 " MPPI_R1 "\n\t"  // yl += rot1(xl) M1
        "v_fmac_f32_dpp %0, %15, %7 " MPPI_R2 "\n\t"   // ya += rot2(xl) H2
        "v_fmac_f32_dpp %1, %15, %13 " MPPI_R2         // yl += rot2(xl) M2
        : "=&v"(ya), "=&v"(yl)
        : "v"(I[0]), "v"(I[1]), "v"(I[2]), "v"(H[0]), "v"(H[1]), "v"(H[2]), "v"(Ht[0]), "v"(Ht[1]), "v"(Ht[2]), "v"(M[0]), "v"(M[1]), "v"(M[2]),
          "v"(xa), "v"(xl));
    return QSV{ya, yl};
}
// X[j] += s * rot_j(y) for one rotated-row block, j = 0, 1, 2 (the rank-one update of the articulated inertia: s = -u_r / d)
__device__ __forceinline__ void qrank1_fused(QF *Ia, QF *Ha, QF *Hta, QF *Ma, QF sn, QF sf, QF ua, QF ul) {
    asm("v_fmac_f32 %0, %12, %14\n\t"                // I0  += sn ua
        "v_fmac_f32 %3, %12, %15\n\t"                // H0  += sn ul
        "v_fmac_f32 %6, %13, %14\n\t"                // Ht0 += sf ua
        "v_fmac_f32 %9, %13, %15\n\t"                // M0  += sf ul
        "v_fmac_f32_dpp %1, %14, %12 " MPPI_R1 "\n\t"   // I1  += rot1(ua) sn
        "v_fmac_f32_dpp %2, %14, %12 " MPPI_R2 "\n\t"   // I2  += rot2(ua) sn
        "v_fmac_f32_dpp %4, %15, %12 " MPPI_R1 "\n\t"   // H1  += rot1(ul) sn
        "v_fmac_f32_dpp %5, %15, %12 " MPPI_R2 "\n\t"   // H2  += rot2(ul) sn
        "v_fmac_f32_dpp %7, %14, %13 " MPPI_R1 "\n\t"   // Ht1 += rot1(ua) sf
        "v_fmac_f32_dpp %8, %14, %13 " MPPI_R2 "\n\t"   // Ht2 += rot2(ua) sf
        "v_fmac_f32_dpp %10, %15, %13 " MPPI_R1 "\n\t"  // M1  += rot1(ul) sf
        "v_fmac_f32_dpp %11, %15, %13 " MPPI_R2         // M2  += rot2(ul) sf
        : "+v"(Ia[0]), "+v"(Ia[1]), "+v"(Ia[2]), "+v"(Ha[0]), "+v"(Ha[1]), "+v"(Ha[2]), "+v"(Hta[0]), "+v"(Hta[1]), "+v"(Hta[2]), "+v"(Ma[0]),
          "+v"(Ma[1]), "+v"(Ma[2])
        : "v"(sn), "v"(sf), "v"(ua), "v"(ul));
}
#else
// (the four UNROTATED products of a 6x6 times a spatial vector pair up as (I0, Ht0) xa + (H0, M0) xl: two packed
// multiply-adds; the register pairs are formed from the rows where they live - the allocator places them, no copies)
__device__ __forceinline__ QSV qmul_fused(const QF *I, const QF *H, const QF *Ht, const QF *M, QF xa, QF xl) {
    QF2 y;
    const QF2 IHt0 = {I[0], Ht[0]}, HM0 = {H[0], M[0]}, x = {xa, xl};
    asm("v_pk_mul_f32 %0, %1, %3 op_sel_hi:[1,0]\n\t"                       // (ya, yl)  = (I0, Ht0) xa
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]"        // (ya, yl) += (H0, M0) xl
        : "=&v"(y)
        : "v"(IHt0), "v"(HM0), "v"(x));
    // the rotated terms read xa, xl through DPP: as halves of the pair the packed instructions above have just read, so that
    // their last write lies at least those two instructions back (tests/test_dpp_hazards.py checks the built library)
    QF ya = y.x, yl = y.y;
    xa = x.x;
    xl = x.y;
    asm(MPPI_LEAD
        "v_fmac_f32_dpp %0, %10, %2 " MPPI_R1 "\n\t"   // ya += rot1(xa) I1
        "v_fmac_f32_dpp %1, %10, %6 " MPPI_R1 "\n\t"   // yl += rot1(xa) Ht1
        "v_fmac_f32_dpp %0, %10, %3 " MPPI_R2 "\n\t"   // ya += rot2(xa) I2
        "v_fmac_f32_dpp %1, %10, %7 " MPPI_R2 "\n\t"   // yl += rot2(xa) Ht2
        "v_fmac_f32_dpp %0, %11, %4 " MPPI_R1 "\n\t"   // ya += rot1(xl) H1
        "v_fmac_f32_dpp %1, %11, %8 " MPPI_R1 "\n\t"   // yl += rot1(xl) M1
        "v_fmac_f32_dpp %0, %11, %5 " MPPI_R2 "\n\t"   // ya += rot2(xl) H2
        "v_fmac_f32_dpp %1, %11, %9 " MPPI_R2            // yl += rot2(xl) M2
        : "+v"(ya), "+v"(yl)
        : "v"(I[1]), "v"(I[2]), "v"(H[1]), "v"(H[2]), "v"(Ht[1]), "v"(Ht[2]), "v"(M[1]), "v"(M[2]), "v"(xa), "v"(xl));
    return QSV{ya, yl};
}
// X[j] += s * rot_j(y) for one rotated-row block, j = 0, 1, 2 (the rank-one update of the articulated inertia: s = -u_r / d)
__device__ __forceinline__ void qrank1_fused(QF *Ia, QF *Ha, QF *Hta, QF *Ma, QF sn, QF sf, QF ua, QF ul) {
    QF2 IHt0 = {Ia[0], Hta[0]}, HM0 = {Ha[0], Ma[0]};
    const QF2 s = {sn, sf}, u = {ua, ul};
    asm("v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"  // (I0, Ht0) += (sn, sf) ua
        "v_pk_fma_f32 %1, %2, %3, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]"        // (H0, M0)  += (sn, sf) ul
        : "+v"(IHt0), "+v"(HM0)
        : "v"(s), "v"(u));
    Ia[0] = IHt0.x; Hta[0] = IHt0.y;
    Ha[0] = HM0.x;  Ma[0] = HM0.y;
    ua = u.x;  // (DPP operands below: the halves the packed instructions have just read, see qmul_fused)
    ul = u.y;
    asm(MPPI_LEAD
        "v_fmac_f32_dpp %0, %10, %8 " MPPI_R1 "\n\t"   // I1  += rot1(ua) sn
        "v_fmac_f32_dpp %1, %10, %8 " MPPI_R2 "\n\t"   // I2  += rot2(ua) sn
        "v_fmac_f32_dpp %2, %11, %8 " MPPI_R1 "\n\t"   // H1  += rot1(ul) sn
        "v_fmac_f32_dpp %3, %11, %8 " MPPI_R2 "\n\t"   // H2  += rot2(ul) sn
        "v_fmac_f32_dpp %4, %10, %9 " MPPI_R1 "\n\t"   // Ht1 += rot1(ua) sf
        "v_fmac_f32_dpp %5, %10, %9 " MPPI_R2 "\n\t"   // Ht2 += rot2(ua) sf
        "v_fmac_f32_dpp %6, %11, %9 " MPPI_R1 "\n\t"   // M1  += rot1(ul) sf
        "v_fmac_f32_dpp %7, %11, %9 " MPPI_R2            // M2  += rot2(ul) sf
        : "+v"(Ia[1]), "+v"(Ia[2]), "+v"(Ha[1]), "+v"(Ha[2]), "+v"(Hta[1]), "+v"(Hta[2]), "+v"(Ma[1]), "+v"(Ma[2])
        : "v"(sn), "v"(sf), "v"(ua), "v"(ul));
}
#endif
// rotated rows of the rigid inertia about the world origin, R Ic R^T + m(|c|^2 1 - c c^T) with h = m c, cw = c:
//   X_j = Tr0 rot_j(R0) + Tr1 rot_j(R1) + Tr2 rot_j(R2) - h rot_j(cw)   (j = 1, 2)
//   X_0 = Tr0 R0 + Tr1 R1 + Tr2 R2 + rot1(h cw) + rot2(h cw)            (the diagonal: |h|^2/m - h_r cw_r is the sum of the
//                                                                         OTHER two products - no cross-lane sum, no cancellation)
__device__ __forceinline__ void qinertia_rows_fused(const QF *Tr, const QF *Rc, QF h, QF cw, QF &I0, QF &I1, QF &I2) {
    QF x0, x1, x2, t;
    asm("v_mul_f32 %3, %10, %11\n\t"                    //  1 t   = h cw
        "v_mul_f32 %0, %4, %7\n\t"                      //  2 x0  = Tr0 R0
        "v_fmac_f32 %0, %5, %8\n\t"                     //  3 x0 += Tr1 R1
        "v_fmac_f32 %0, %6, %9\n\t"                     //  4 x0 += Tr2 R2
        "v_mul_f32_dpp %1, %7, %4 " MPPI_R1 "\n\t"      //  5 x1  = rot1(R0) Tr0
        "v_mul_f32_dpp %2, %7, %4 " MPPI_R2 "\n\t"      //  6 x2  = rot2(R0) Tr0
        "v_fmac_f32_dpp %1, %8, %5 " MPPI_R1 "\n\t"
        "v_fmac_f32_dpp %2, %8, %5 " MPPI_R2 "\n\t"
        "v_fmac_f32_dpp %1, %9, %6 " MPPI_R1 "\n\t"
        "v_fmac_f32_dpp %2, %9, %6 " MPPI_R2 "\n\t"
        "v_fmac_f32_dpp %1, %11, -%10 " MPPI_R1 "\n\t"  // 11 x1 -= rot1(cw) h
        "v_fmac_f32_dpp %2, %11, -%10 " MPPI_R2 "\n\t"  // 12 x2 -= rot2(cw) h
        "v_add_f32_dpp %0, %3, %0 " MPPI_R1 "\n\t"      // 13 x0 += rot1(t)          (t written at 1)
        "v_add_f32_dpp %0, %3, %0 " MPPI_R2               // 14 x0 += rot2(t)
        : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(t)
        : "v"(Tr[0]), "v"(Tr[1]), "v"(Tr[2]), "v"(Rc[0]), "v"(Rc[1]), "v"(Rc[2]), "v"(h), "v"(cw));
    I0 = x0;
    I1 = x1;
    I2 = x2;
}
// v = v_parent + S qd and the velocity-product bias c = v_parent x (S qd) of one joint (pass 1 of the articulated-body solve):
// cross products as  rot1(a rot1(b) - rot1(a) b), the two of c.l summed before their common rotation
__device__ __forceinline__ void qvel_bias_fused(QF vpa, QF vpl, QF Sa, QF Sl, QF qd, QF &va, QF &vl, QF &cba, QF &cbl) {
    QF sja, sjl, ta, tl;
    asm("v_mul_f32 %4, %12, %10\n\t"                    //  1 sja = qd Sa
        "v_mul_f32 %5, %12, %11\n\t"                    //  2 sjl = qd Sl
        "v_add_f32 %0, %8, %4\n\t"                      //  3 va  = vpa + sja
        "v_mul_f32_dpp %6, %4, %8 " MPPI_R1 "\n\t"      //  4 ta  = rot1(sja) vpa
        "v_fmac_f32_dpp %6, %8, -%4 " MPPI_R1 "\n\t"    //  5 ta -= rot1(vpa) sja
        "v_mul_f32_dpp %7, %5, %8 " MPPI_R1 "\n\t"      //  6 tl  = rot1(sjl) vpa
        "v_fmac_f32_dpp %7, %8, -%5 " MPPI_R1 "\n\t"    //  7 tl -= rot1(vpa) sjl
        "v_fmac_f32_dpp %7, %4, %9 " MPPI_R1 "\n\t"     //  8 tl += rot1(sja) vpl
        "v_fmac_f32_dpp %7, %9, -%4 " MPPI_R1 "\n\t"    //  9 tl -= rot1(vpl) sja
        "v_mov_b32_dpp %2, %6 " MPPI_R1 "\n\t"          // 10 cba = rot1(ta)      (ta written at 5)
        "v_add_f32 %1, %9, %5\n\t"                      // 11 vl  = vpl + sjl
        "v_mov_b32_dpp %3, %7 " MPPI_R1                 // 12 cbl = rot1(tl)      (tl written at 9)
        : "=&v"(va), "=&v"(vl), "=&v"(cba), "=&v"(cbl), "=&v"(sja), "=&v"(sjl), "=&v"(ta), "=&v"(tl)
        : "v"(vpa), "v"(vpl), "v"(Sa), "v"(Sl), "v"(qd));
}
// velocity-product force of a rigid body about the world origin, pA = v x* (I v):
//   n = I w + h x vl,  f = m vl + w x h,  pA = (w x n + vl x f, w x f)      (I in rotated rows I0, I1, I2; h = m c)
__device__ __forceinline__ void qbias_force_fused(QF I0, QF I1, QF I2, QF h, QF m, QF w, QF vl, QF &pAa, QF &pAl) {
    QF n, f, t1, t2, t3, t4;
    asm("v_mul_f32 %2, %8, %13\n\t"                     //  1 n   = I0 w
        "v_mul_f32 %3, %12, %14\n\t"                    //  2 f   = m vl
        "v_mul_f32_dpp %4, %14, %11 " MPPI_R1 "\n\t"    //  3 t1  = rot1(vl) h
        "v_fmac_f32_dpp %4, %11, -%14 " MPPI_R1 "\n\t"  //  4 t1 -= rot1(h) vl        t1 = (h x vl) before its rotation
        "v_mul_f32_dpp %5, %11, %13 " MPPI_R1 "\n\t"    //  5 t2  = rot1(h) w
        "v_fmac_f32_dpp %5, %13, -%11 " MPPI_R1 "\n\t"  //  6 t2 -= rot1(w) h         t2 = (w x h) before its rotation
        "v_fmac_f32_dpp %2, %13, %9 " MPPI_R1 "\n\t"    //  7 n  += rot1(w) I1
        "v_fmac_f32_dpp %2, %13, %10 " MPPI_R2 "\n\t"   //  8 n  += rot2(w) I2
        "v_add_f32_dpp %2, %4, %2 " MPPI_R1 "\n\t"      //  9 n  += rot1(t1)          (t1 written at 4)
        "v_add_f32_dpp %3, %5, %3 " MPPI_R1 "\n\t"      // 10 f  += rot1(t2)          (t2 written at 6)
        "v_mul_f32_dpp %6, %13, -%2 " MPPI_R1 "\n\t"    // 11 t3  = -rot1(w) n
        "v_mul_f32_dpp %7, %13, -%3 " MPPI_R1 "\n\t"    // 12 t4  = -rot1(w) f
        "v_fmac_f32_dpp %6, %2, %13 " MPPI_R1 "\n\t"    // 13 t3 += rot1(n) w         (n written at 9)
        "v_fmac_f32_dpp %7, %3, %13 " MPPI_R1 "\n\t"    // 14 t4 += rot1(f) w         (f written at 10)
        "v_fmac_f32_dpp %6, %3, %14 " MPPI_R1 "\n\t"    // 15 t3 += rot1(f) vl
        "v_fmac_f32_dpp %6, %14, -%3 " MPPI_R1 "\n\t"   // 16 t3 -= rot1(vl) f
        "v_mov_b32_dpp %1, %7 " MPPI_R1 "\n\t"          // 17 pAl = rot1(t4)          (t4 written at 14)
        "s_nop 0\n\t"                                   //    (t3 written at 16: one more wait state)
        "v_mov_b32_dpp %0, %6 " MPPI_R1                 // 18 pAa = rot1(t3)
        : "=&v"(pAa), "=&v"(pAl), "=&v"(n), "=&v"(f), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4)
        : "v"(I0), "v"(I1), "v"(I2), "v"(h), "v"(m), "v"(w), "v"(vl));
}
// One joint of the inward pass after U = IA S:  d = kdh + S.U,  u = tau - S.pA,  1/d,  W = -U/d.  The 6-dots are summed over
// the quad from broadcasts, in every lane alike ((c + t0) + t1) + t2 - replicated scalars stay bit-identical across the quad -
// and interleaved so that no DPP read follows the write of its operand (a dot on its own is mul, fmac, WAIT, DPP, DPP, WAIT,
// DPP).  1/d is the hardware reciprocal (1 ulp; the Newton step that used to follow bought nothing the 1e-4 cost tolerance
// sees); it sits two instructions before the end of the block, so that whatever reads it next is past the wait state a
// transcendental's result needs (gfx940+: no VALU read in the issue slot right behind it - the compiler does not see into here).
#define MPPI_B(k) "quad_perm:[" #k "," #k "," #k "," #k "] row_mask:0xf bank_mask:0xf bound_ctrl:1"
__device__ __forceinline__ void qjoint_fused(QF Sa, QF Sl, QF Ua, QF Ul, QF pa, QF pl, QF kdh, QF tau, QF &u, QF &invd, QF &Wa, QF &Wl) {
    QF d, t1, t2;
    asm("v_mul_f32 %5, %7, %9\n\t"                        //  1 t1  = Sa Ua
        "v_fmac_f32 %5, %8, %10\n\t"                      //  2 t1 += Sl Ul
        "v_mul_f32 %6, %7, %11\n\t"                       //  3 t2  = Sa pa
        "v_fmac_f32 %6, %8, %12\n\t"                      //  4 t2 += Sl pl
        "v_add_f32_dpp %4, %5, %13 " MPPI_B(0) "\n\t"     //  5 d   = t1[0] + kdh      (t1 written at 2)
        "v_add_f32_dpp %4, %5, %4 " MPPI_B(1) "\n\t"      //  6 d  += t1[1]
        "v_subrev_f32_dpp %0, %6, %14 " MPPI_B(0) "\n\t"  //  7 u   = tau - t2[0]      (t2 written at 4)
        "v_add_f32_dpp %4, %5, %4 " MPPI_B(2) "\n\t"      //  8 d  += t1[2]
        "v_subrev_f32_dpp %0, %6, %0 " MPPI_B(1) "\n\t"   //  9 u  -= t2[1]
        "v_rcp_f32 %1, %4\n\t"                            // 10 1/d
        "v_subrev_f32_dpp %0, %6, %0 " MPPI_B(2) "\n\t"   // 11 u  -= t2[2]
        "v_mul_f32 %2, %9, -%1\n\t"                       // 12 Wa  = -Ua / d          (1/d written at 10)
        "v_mul_f32 %3, %10, -%1"                           // 13 Wl  = -Ul / d
        : "=&v"(u), "=&v"(invd), "=&v"(Wa), "=&v"(Wl), "=&v"(d), "=&v"(t1), "=&v"(t2)
        : "v"(Sa), "v"(Sl), "v"(Ua), "v"(Ul), "v"(pa), "v"(pl), "v"(kdh), "v"(tau));
}
// ... and what goes on to the parent:  k = (u - U.c)/d,  pa = pA + IA c + k U  (IA c from the caller); the two sums pA + IA c
// fill the wait between the dot's products and their first DPP read
__device__ __forceinline__ void qbias_to_parent_fused(QF Ua, QF Ul, QF ca, QF cl, QF u, QF invd, QF pAa, QF pAl, QF Aca, QF Acl, QF &k, QF &paa,
                                                      QF &pal) {
    QF t;
    asm("v_mul_f32 %3, %4, %6\n\t"                        //  1 t    = Ua ca
        "v_fmac_f32 %3, %5, %7\n\t"                       //  2 t   += Ul cl
        "v_add_f32 %1, %10, %12\n\t"                      //  3 paa  = pAa + (IA c).a
        "v_add_f32 %2, %11, %13\n\t"                      //  4 pal  = pAl + (IA c).l
        "v_subrev_f32_dpp %0, %3, %8 " MPPI_B(0) "\n\t"   //  5 k    = u - t[0]         (t written at 2)
        "v_subrev_f32_dpp %0, %3, %0 " MPPI_B(1) "\n\t"   //  6 k   -= t[1]
        "v_subrev_f32_dpp %0, %3, %0 " MPPI_B(2) "\n\t"   //  7 k   -= t[2]
        "v_mul_f32 %0, %0, %9\n\t"                        //  8 k   *= 1/d
        "v_fmac_f32 %1, %0, %4\n\t"                       //  9 paa += k Ua
        "v_fmac_f32 %2, %0, %5"                            // 10 pal += k Ul
        : "=&v"(k), "=&v"(paa), "=&v"(pal), "=&v"(t)
        : "v"(Ua), "v"(Ul), "v"(ca), "v"(cl), "v"(u), "v"(invd), "v"(pAa), "v"(pAl), "v"(Aca), "v"(Acl));
}
// outward pass of one joint: qdd = k + W . a_parent,  a = (a_parent + c) + qdd S; the two sums a_parent + c fill the wait
// between the dot's products and their first DPP read
__device__ __forceinline__ void qoutward_fused(QF Wa, QF Wl, QF apa, QF apl, QF ca, QF cl, QF Sa, QF Sl, QF k, QF &dd, QF &aa, QF &al) {
    QF t;
    asm("v_mul_f32 %3, %4, %6\n\t"                       //  1 t   = Wa apa
        "v_fmac_f32 %3, %5, %7\n\t"                      //  2 t  += Wl apl
        "v_add_f32 %1, %6, %8\n\t"                       //  3 aa  = apa + ca
        "v_add_f32 %2, %7, %9\n\t"                       //  4 al  = apl + cl
        "v_add_f32_dpp %0, %3, %12 " MPPI_B(0) "\n\t"    //  5 dd  = t[0] + k         (t written at 2)
        "v_add_f32_dpp %0, %3, %0 " MPPI_B(1) "\n\t"     //  6 dd += t[1]
        "v_add_f32_dpp %0, %3, %0 " MPPI_B(2) "\n\t"     //  7 dd += t[2]
        "v_fmac_f32 %1, %0, %10\n\t"                     //  8 aa += dd Sa
        "v_fmac_f32 %2, %0, %11"                          //  9 al += dd Sl
        : "=&v"(dd), "=&v"(aa), "=&v"(al), "=&v"(t)
        : "v"(Wa), "v"(Wl), "v"(apa), "v"(apl), "v"(ca), "v"(cl), "v"(Sa), "v"(Sl), "v"(k));
}
#endif

// y = A x for the 6x6: rotated rows meet rotated copies of the distributed vector
MPPI_HD QSV qmul(const QAI &A, const QSV &x) {
#if defined(MPPI_DPP_FMAC)
    return qmul_fused(A.I, A.H, A.Ht, A.M, x.a, x.l);
#endif
    const QF a1 = rot1(x.a), a2 = rot2(x.a), l1 = rot1(x.l), l2 = rot2(x.l);
    QSV y;
    y.a = A.I[0] * x.a + A.I[1] * a1 + A.I[2] * a2 + A.H[0] * x.l + A.H[1] * l1 + A.H[2] * l2;
    y.l = A.Ht[0] * x.a + A.Ht[1] * a1 + A.Ht[2] * a2 + A.M[0] * x.l + A.M[1] * l1 + A.M[2] * l2;
    return y;
}
MPPI_HD QF qdot6(const QSV &p, const QSV &q) { return qsum(p.a * q.a + p.l * q.l); }

// JT: joint types known at compile time (0: every joint is revolute, -1: read per joint from the model).  The kernel picks
// the instantiation with ONE wave-uniform branch; inside, an all-revolute arm then has no per-joint type test at all
// (each was a compare on a VGPR-resident uniform value plus an exec-mask region or a select).
// World poses of the moving bodies, kept as the pairs the kinematics produce and consume: (column 0, column 1) of R and
// (column 2, position).  rot() / pos() hand out the halves.
template <class T, int JT = -1>
struct QPose {
    QF2 R01[T::NB ? T::NB : 1], R2p[T::NB ? T::NB : 1];
    int jt[T::NB ? T::NB : 1];
    QF2 Rb01, Rb2p;
    MPPI_HD bool revolute(int i) const {
        if constexpr (JT == 0) return true;
        else return jt[i] == 0;
    }
    MPPI_HD QM3 rot(int i) const { return QM3{{R01[i].x, R01[i].y, R2p[i].x}}; }
    MPPI_HD QF pos(int i) const { return R2p[i].y; }
    MPPI_HD QM3 rot_base() const { return QM3{{Rb01.x, Rb01.y, Rb2p.x}}; }
    MPPI_HD QF pos_base() const { return Rb2p.y; }
    MPPI_HD void set_base(const QM3 &R, QF p) {
        Rb01.x = R.c[0]; Rb01.y = R.c[1];
        Rb2p.x = R.c[2]; Rb2p.y = p;
    }
};

// first moment h = R hb + m p and Tr = R Ic of a body posed at (R, p), the inputs of its world-frame rigid inertia: thirteen
// products-and-sums, on the device as six packed multiply-adds over the constant pairs of BodyK1 plus one for m p
MPPI_HD void qmoments(QF2 R01, QF2 R2p, const BodyK1 &b, QF &h, QF *Tr) {
#if defined(MPPI_DPP_FMAC) && !defined(MPPI_NO_PK_MOMENTS)
    const QF2 A0 = {b.hI[0], b.hI[1]}, A1 = {b.hI[2], b.hI[3]}, A2 = {b.hI[4], b.hI[5]}, B0 = {b.II[0], b.II[1]}, B1 = {b.II[2], b.II[3]},
              B2 = {b.II[4], b.II[5]};
    QF2 X, Y;
    asm("v_pk_mul_f32 %0, %2, %4 op_sel_hi:[0,1]\n\t"                          // (h, Tr0)    = c0 (hb0, Ic0)
        "v_pk_mul_f32 %1, %2, %7 op_sel_hi:[0,1]\n\t"                          // (Tr1, Tr2)  = c0 (Ic1, Ic2)
        "v_pk_fma_f32 %0, %2, %5, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"     // (h, Tr0)   += c1 (hb1, Ic1)
        "v_pk_fma_f32 %1, %2, %8, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"     // (Tr1, Tr2) += c1 (Ic3, Ic4)
        "v_pk_fma_f32 %0, %3, %6, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"     // (h, Tr0)   += c2 (hb2, Ic2)
        "v_pk_fma_f32 %1, %3, %9, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]"           // (Tr1, Tr2) += c2 (Ic4, Ic5)
        : "=&v"(X), "=&v"(Y)
        : "v"(R01), "v"(R2p), "v"(A0), "v"(A1), "v"(A2), "v"(B0), "v"(B1), "v"(B2));
    h = X.x + b.m * R2p.y;
    Tr[0] = X.y;
    Tr[1] = Y.x;
    Tr[2] = Y.y;
#else
    const QF c0 = R01.x, c1 = R01.y, c2 = R2p.x;
    h = c0 * b.hb(0) + c1 * b.hb(1) + c2 * b.hb(2) + b.m * R2p.y;
    Tr[0] = c0 * b.Ic(0) + c1 * b.Ic(1) + c2 * b.Ic(2);
    Tr[1] = c0 * b.Ic(1) + c1 * b.Ic(3) + c2 * b.Ic(4);
    Tr[2] = c0 * b.Ic(2) + c1 * b.Ic(4) + c2 * b.Ic(5);
#endif
}

// all kinematic blocks are requested up front (LDS returns in order, so body i only waits for its own
// block while the later ones stream in behind the arithmetic)
template <class T, class M>
MPPI_HD void quad_fk_blocks(M &m, BodyK0 *blk) {
    static_for<0, T::NB>([&](auto ic) MPPI_LAMBDA { blk[ic] = load_block<BodyK0>(m.b[ic].k0); });
}
template <class T, int JT>
MPPI_HD void quad_fk_from(const BodyK0 *blk, const QF *q, QPose<T, JT> &P);
template <class T, class M, int JT>
MPPI_HD void quad_fk(M &m, const QF *q, QPose<T, JT> &P) {
    BodyK0 blk[T::NB ? T::NB : 1];
    quad_fk_blocks<T>(m, blk);
    quad_fk_from<T, JT>(blk, q, P);
}
template <class T, int JT>
MPPI_HD void quad_fk_from(const BodyK0 *blk, const QF *q, QPose<T, JT> &P) {
    static_for<0, T::NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const BodyK0 &b = blk[i];
        P.jt[i] = b.jtype;
        const QF2 Rp01 = par < 0 ? P.Rb01 : P.R01[par < 0 ? 0 : par], Rp2p = par < 0 ? P.Rb2p : P.R2p[par < 0 ? 0 : par];
        const bool rev = JT == 0 || b.jtype == 0;
        QF sn = qrep(0.f), cn = qrep(1.f);  // prismatic: the identity rotation
        if (rev) qsincos(q[i], sn, cn);
        QF2 R01, RT2p;  // columns 0, 1 of R_parent Rt Rz(q) | column 2 and R_parent pt
#if defined(MPPI_DPP_FMAC) && !defined(MPPI_NO_PK_FK)
        // six packed multiply-adds for the twelve products-and-sums of [R_parent Rt | R_parent pt] - the parent's column k is
        // broadcast to both halves by op_sel, the constants are the 3x4 block's row pairs - and two for the rotation about the
        // joint's z, (c0', c1') = cos q (c0, c1) + sin q (c1, -c0).
        // HAZARD (gfx940+): a VALU instruction must not read the result of a transcendental (v_sin, v_cos, v_rcp ...) in the
        // very next issue slot.  The compiler pads its own code, not inline assembly: (cos, sin) is an operand of THIS block, so
        // it is complete before the block starts and first read six instructions in (tools/check_dpp_hazards.py looks for
        // this as well; a separate rotation block right behind v_sin / v_cos read a stale sine).
        const QF2 K01 = {b.T[0], b.T[1]}, K23 = {b.T[2], b.T[3]}, K45 = {b.T[4], b.T[5]}, K67 = {b.T[6], b.T[7]}, K89 = {b.T[8], b.T[9]},
                  Kab = {b.T[10], b.T[11]}, cs = {cn, sn};
        QF2 RT01;
        asm("v_pk_mul_f32 %0, %3, %5 op_sel_hi:[0,1]\n\t"                                      // RT01  = c0 (Rt00, Rt01)
            "v_pk_mul_f32 %1, %3, %6 op_sel_hi:[0,1]\n\t"                                      // RT2p  = c0 (Rt02, pt0)
            "v_pk_fma_f32 %0, %3, %7, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                 // RT01 += c1 (Rt10, Rt11)
            "v_pk_fma_f32 %1, %3, %8, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"                 // RT2p += c1 (Rt12, pt1)
            "v_pk_fma_f32 %0, %4, %9, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"                 // RT01 += c2 (Rt20, Rt21)
            "v_pk_fma_f32 %1, %4, %10, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"                // RT2p += c2 (Rt22, pt2)
            "v_pk_mul_f32 %2, %0, %11 op_sel_hi:[1,0]\n\t"                                     // R01   = (c0, c1) cos
            "v_pk_fma_f32 %2, %0, %11, %2 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]"      // R01  += (c1 sin, -c0 sin)
            : "=&v"(RT01), "=&v"(RT2p), "=&v"(R01)
            : "v"(Rp01), "v"(Rp2p), "v"(K01), "v"(K23), "v"(K45), "v"(K67), "v"(K89), "v"(Kab), "v"(cs));
#else
        const QF c0 = Rp01.x, c1 = Rp01.y, c2 = Rp2p.x;
        const QF r0 = c0 * b.rt(0) + c1 * b.rt(3) + c2 * b.rt(6);
        const QF r1 = c0 * b.rt(1) + c1 * b.rt(4) + c2 * b.rt(7);
        RT2p.x = c0 * b.rt(2) + c1 * b.rt(5) + c2 * b.rt(8);
        RT2p.y = c0 * b.pt(0) + c1 * b.pt(1) + c2 * b.pt(2);
        R01.x = cn * r0 + sn * r1;
        R01.y = cn * r1 - sn * r0;
#endif
        P.R01[i] = R01;
        P.R2p[i].x = RT2p.x;
        const QF pw = Rp2p.y + RT2p.y;
        P.R2p[i].y = rev ? pw : pw + q[i] * RT2p.x;  // prismatic along the joint's z
    });
}

template <class T, int i, int JT>
MPPI_HD QSV quad_subspace(const QPose<T, JT> &P) {
    const QF az = P.R2p[i].x;  // third column of R: component r lives in lane r's row
    if (P.revolute(i)) return {az, qcross(P.pos(i), az)};
    return {qrep(0.f), az};
}

// Articulated-body solve, quad-parallel.  tau/kdh/qd/qdd are replicated scalars (same in all lanes of a quad).
struct JointLimits {  // per-joint drive limits, cached from block 1 while the solve has it in registers (+inf: none)
    float effort, vmax;
};

template <class T, class M, int JT>
MPPI_HD void quad_aba(M &m, const QPose<T, JT> &P, const QF *qd, const QF *tau_exp, const QF *kdh, QF *qdd, JointLimits *lim) {
    constexpr int NB = T::NB;
    QSV v[NB], W[NB], pacc[NB], cb[NB];
    QF Sl[NB];  // linear part of the joint subspace (the angular part is the third column of R)
    QAI acc[NB];
    QF kk[NB];
    bool has_acc[NB];
    const QF zero = qrep(0.f);
    // pass 1: velocities, and the two quantities both later sweeps need: S_i and c_i = v_parent x (S_i qd_i)
    // (the linear parts p x az of all joints first, products before their common rotation: the rotation reads its operand
    // through DPP two wait states after the write - with the other joints' products in between nobody waits)
    QF St[NB];
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        const QF az = P.R2p[i].x, p = P.pos(i);
        St[i] = P.revolute(i) ? p * rot1(az) - rot1(p) * az : az;
    });
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA { Sl[ic] = P.revolute(ic) ? rot1(St[ic]) : St[ic]; });
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const QSV S = {P.revolute(i) ? P.R2p[i].x : zero, Sl[i]};
        const QSV sj = {qd[i] * S.a, qd[i] * S.l};
        if constexpr (par < 0) {
            v[i] = sj;
            cb[i] = {zero, zero};
        } else {
            const QSV vp = v[par < 0 ? 0 : par];
#if defined(MPPI_DPP_FMAC)
            qvel_bias_fused(vp.a, vp.l, S.a, S.l, qd[i], v[i].a, v[i].l, cb[i].a, cb[i].l);
#else
            v[i] = {vp.a + sj.a, vp.l + sj.l};
            cb[i] = {qcross(vp.a, sj.a), qcross(vp.a, sj.l) + qcross(vp.l, sj.a)};
#endif
        }
        has_acc[i] = false;
    });
    BodyK1 blk[NB];  // requested leaf-first, in the order the backward sweep consumes them
    static_rfor<0, NB>([&](auto ic) MPPI_LAMBDA { blk[ic] = load_block<BodyK1>(m.b[ic].k1); });
    static_rfor<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const BodyK1 &b = blk[i];
        lim[i] = {b.effort, b.vmax};
        const QM3 R = P.rot(i);
        const QSV S = {P.revolute(i) ? R.c[2] : zero, Sl[i]};
        // rigid inertia about the world origin: I_O = R Ic R^T + m(|cw|^2 1 - cw cw^T), h = m cw
        QF h, Tr[3];
        qmoments(P.R01[i], P.R2p[i], b, h, Tr);
        const QF cw = b.invm * h;
        const QF h1 = rot1(h), h2 = rot2(h);
        QAI A;
#if defined(MPPI_DPP_FMAC)
        qinertia_rows_fused(Tr, R.c, h, cw, A.I[0], A.I[1], A.I[2]);
#else
        const QF hh = qsum(h * cw);
        A.I[0] = Tr[0] * R.c[0] + Tr[1] * R.c[1] + Tr[2] * R.c[2] + hh - h * cw;
        A.I[1] = Tr[0] * rot1(R.c[0]) + Tr[1] * rot1(R.c[1]) + Tr[2] * rot1(R.c[2]) - h * rot1(cw);
        A.I[2] = Tr[0] * rot2(R.c[0]) + Tr[1] * rot2(R.c[1]) + Tr[2] * rot2(R.c[2]) - h * rot2(cw);
#endif
        // skew(h) in rotated rows is (0, -h2, h1), its transpose the negative, the mass block (m, 0, 0): on top of the children's
        // articulated inertias where there are any (written out per entry - the sums with the structural zeros are not
        // the compiler's to drop without fast-math)
        if (has_acc[i]) {
            const QAI &C = acc[i];
            A.H[0] = C.H[0];        A.H[1] = C.H[1] - h2;   A.H[2] = C.H[2] + h1;
            A.Ht[0] = C.Ht[0];      A.Ht[1] = C.Ht[1] + h2; A.Ht[2] = C.Ht[2] - h1;
            A.M[0] = C.M[0] + b.m;  A.M[1] = C.M[1];        A.M[2] = C.M[2];
        } else {
            A.H[0] = qrep(0.f); A.H[1] = -h2;       A.H[2] = h1;
            A.Ht[0] = A.H[0];  A.Ht[1] = h2;       A.Ht[2] = -h1;
            A.M[0] = qrep(b.m); A.M[1] = A.H[0];   A.M[2] = A.H[0];
        }
        // bias force v x* (I v)
        const QF w = v[i].a, vl = v[i].l;
#if defined(MPPI_DPP_FMAC)
        QSV pA;
        qbias_force_fused(A.I[0], A.I[1], A.I[2], h, qrep(b.m), w, vl, pA.a, pA.l);
#else
        const QF n = A.I[0] * w + A.I[1] * rot1(w) + A.I[2] * rot2(w) + qcross(h, vl);
        const QF f = b.m * vl + qcross(w, h);
        QSV pA = {qcross(w, n) + qcross(vl, f), qcross(w, f)};
#endif
        if (has_acc[i]) {  // (the velocity-product force above is the RIGID body's: its inertia rows meet the children's only now)
            for (int j = 0; j < 3; j++) A.I[j] += acc[i].I[j];
            pA = {pA.a + pacc[i].a, pA.l + pacc[i].l};
        }
        const QSV Ui = qmul(A, S);
#if defined(MPPI_DPP_FMAC)
        QF ui, invd;
        qjoint_fused(S.a, S.l, Ui.a, Ui.l, pA.a, pA.l, kdh[i], tau_exp[i], ui, invd, W[i].a, W[i].l);
#else
        const QF invd = qrcp(qdot6(S, Ui) + kdh[i]), ui = tau_exp[i] - qdot6(S, pA);
        // what the outward pass needs of this joint: qdd_i = k_i + W_i . a_parent with W = -U/d and k = (u - U.c)/d
        // (the parent's acceleration enters through one dot; c_i is already folded into k_i here)
        const QF ninvd = -invd;
        W[i] = {Ui.a * ninvd, Ui.l * ninvd};
#endif
        if constexpr (par < 0) kk[i] = ui * invd;
        if constexpr (par >= 0) {
            const QSV c = cb[i];
            const QSV Ac = qmul(A, c);
#if defined(MPPI_DPP_FMAC)
            QSV pa;
            qbias_to_parent_fused(Ui.a, Ui.l, c.a, c.l, ui, invd, pA.a, pA.l, Ac.a, Ac.l, kk[i], pa.a, pa.l);
#else
            const QF k = (ui - qdot6(Ui, c)) * invd;
            kk[i] = k;
            const QSV pa = {pA.a + Ac.a + k * Ui.a, pA.l + Ac.l + k * Ui.l};
#endif
            // Ia = IA - U U^T / d  (rotated rows: X[r][(r+j)%3] += W_r * rot_j(U))
#if defined(MPPI_DPP_FMAC)
            qrank1_fused(A.I, A.H, A.Ht, A.M, W[i].a, W[i].l, Ui.a, Ui.l);
#else
            const QF un = -W[i].a, uf = -W[i].l;
            const QF n1 = rot1(Ui.a), n2 = rot2(Ui.a), f1 = rot1(Ui.l), f2 = rot2(Ui.l);
            A.I[0] -= un * Ui.a; A.I[1] -= un * n1; A.I[2] -= un * n2;
            A.H[0] -= un * Ui.l; A.H[1] -= un * f1; A.H[2] -= un * f2;
            A.Ht[0] -= uf * Ui.a; A.Ht[1] -= uf * n1; A.Ht[2] -= uf * n2;
            A.M[0] -= uf * Ui.l; A.M[1] -= uf * f1; A.M[2] -= uf * f2;
#endif
            constexpr int pj = par < 0 ? 0 : par;
            if (has_acc[pj]) {
                for (int j = 0; j < 3; j++) { acc[pj].I[j] += A.I[j]; acc[pj].H[j] += A.H[j]; acc[pj].Ht[j] += A.Ht[j]; acc[pj].M[j] += A.M[j]; }
                pacc[pj] = {pacc[pj].a + pa.a, pacc[pj].l + pa.l};
            } else {
                acc[pj] = A;
                pacc[pj] = pa;
                has_acc[pj] = true;
            }
        }
    });
    QSV a[NB];
    QSV a0 = {zero, zero};
    if (m.gravity_on) a0.l = qsel(-m.g[0], -m.g[1], -m.g[2]);
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const QSV S = {P.revolute(i) ? P.R2p[i].x : zero, Sl[i]};
        const QSV apar = par >= 0 ? a[par < 0 ? 0 : par] : a0;
#if defined(MPPI_DPP_FMAC)
        qoutward_fused(W[i].a, W[i].l, apar.a, apar.l, cb[i].a, cb[i].l, S.a, S.l, kk[i], qdd[i], a[i].a, a[i].l);  // (root: c = 0)
#else
        const QF dd = kk[i] + qdot6(W[i], apar);
        qdd[i] = dd;
        QSV ap = a0;
        if constexpr (par >= 0) ap = {apar.a + cb[i].a, apar.l + cb[i].l};
        a[i] = {ap.a + dd * S.a, ap.l + dd * S.l};
#endif
    });
}

// How a step solves the articulated body: the quad layout's solve above, or the octet layout's (mppi_oct.hpp OctAba: one
// sample per two quads) - the step / rollout code around the solve is the same for both.
struct QuadAba {
    static constexpr bool kFusedLimitCheck = false;
    template <class T, class M, int JT>
    MPPI_HD void aba(M &m, const QPose<T, JT> &P, const QF *qd, const QF *tau_exp, const QF *kdh, QF *qdd, JointLimits *lim) const {
        quad_aba<T>(m, P, qd, tau_exp, kdh, qdd, lim);
    }
};

// base pose of the (fixed) robot from its root row, distributed over the quad
template <class T, class M, int JT>
MPPI_HD void quad_base(M &m, const float *root, QPose<T, JT> &P) {
    const float *rs = root + 13 * m.robot_actor;
    const M3 R = quat_to_R(rs + 3);
    QM3 Rb;
    for (int c = 0; c < 3; c++) Rb.c[c] = qsel(R.a[c], R.a[3 + c], R.a[6 + c]);
    P.set_base(Rb, qsel(rs[0], rs[1], rs[2]));
}

// One simulator step.  P must hold the forward kinematics of q on entry (base pose included) and holds the
// forward kinematics of the NEW q on exit: the pose computed for the cost / next step is never recomputed.
template <class T, class M, int JT, class AB = QuadAba>
MPPI_HD void quad_step(M &m0, QPose<T, JT> &P, QF *q, QF *qd, const QF *target, const AB &ab = AB{}) {
    constexpr int NB = T::NB;
    M *mp = &m0;
    // position mode (isaacgym_wrapper.py:571-572): apply_robot_cmd overwrites the DOF state with the command.  It lives in the
    // GENERIC instantiation only (JT != 0; mppi_pack.hpp clears all_revolute for a position-driven robot): the all-revolute
    // specialisation is the metric's instruction stream and carries no trace of it (measured: as a run-time branch of the
    // common code it cost the headline kernel 3 %, 0.1203 -> 0.1240 ms)
    if constexpr (JT != 0)
        if (m0.drive_mode == kDrivePosition) {
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA { q[ic] = target[ic]; qd[ic] = qrep(0.f); });
            quad_fk<T>(m0, q, P);
        }
    for (int s = 0; s < m0.substeps; s++) {
        M &m = *launder(mp);
        const float h = m.h, inv_h = frcp(h);
        float kd = m.kd;
        QF tau[NB], kdh[NB], qdd[NB];
        JointLimits lim[NB];
        // joint drives (isaacgym_wrapper.py:491-507): velocity mode tau = kd (target - qd), effort mode tau = target - kd qd,
        // both with the implicit damping kd h qdd inside the solve.  ONE uniform branch picks the mode (as selects it was four
        // instructions per joint and substep).  Position mode (generic instantiation): tau = kp (target - q) - (kd + h kp) qd,
        // the spring at the end-of-substep position (mppi_device.hpp step)
        bool driven = false;
        if constexpr (JT != 0)
            if (m.drive_mode == kDrivePosition) {
                const float kp = m.kp;
                kd += h * kp;
                static_for<0, NB>([&](auto ic) MPPI_LAMBDA { tau[ic] = kp * (target[ic] - q[ic]) - kd * qd[ic]; });
                driven = true;
            }
        if (!driven) {
            if (m.drive_mode == kDriveVelocity) {
                static_for<0, NB>([&](auto ic) MPPI_LAMBDA { tau[ic] = kd * (target[ic] - qd[ic]); });
            } else {
                static_for<0, NB>([&](auto ic) MPPI_LAMBDA { tau[ic] = target[ic] - kd * qd[ic]; });
            }
        }
        const QF kdhq = qrep(kd * h);
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA { kdh[ic] = kdhq; });
        // URDF effort limits: a drive whose torque tau - kd h qdd leaves [-effort, effort] is held at the bound and the step is
        // solved again without its damping.  The test is one running maximum of |torque| - effort over the joints and one
        // branch (no limit: effort = +inf, the excess is -inf); the selects live inside the rare branch.
        QF tt[NB];
        QF excess = qrep(-INFINITY);
        if constexpr (AB::kFusedLimitCheck) {   // (octet layout: the test rides in the wait slots of the solve's outward pass)
            ab.template aba_checked<T>(m, P, qd, tau, kdh, qdd, lim, tt, excess);
        } else {
            ab.template aba<T>(m, P, qd, tau, kdh, qdd, lim);
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
                constexpr int i = ic;
                tt[i] = tau[i] - kdhq * qdd[i];
                excess = qmax(excess, qabs(tt[i]) - qrep(lim[i].effort));
            });
        }
        if (qany_gt(excess, qrep(0.f))) {
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
                constexpr int i = ic;
                const QF eff = qrep(lim[i].effort);
                const bool sat = qany_gt(qabs(tt[i]), eff);
                tau[i] = sat ? qwhere_gt(tt[i], qrep(0.f), eff, -eff) : tau[i];
                kdh[i] = sat ? qrep(0.f) : kdh[i];
            });
            ab.template aba<T>(*launder(mp), P, qd, tau, kdh, qdd, lim);
        }
        // the kinematic blocks of the NEXT pose are requested here: they carry the joint ranges the integration needs, and
        // their round trip runs under the integration's arithmetic
        BodyK0 blk0[NB];
        quad_fk_blocks<T>(*launder(mp), blk0);
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            const JointLimits b = lim[i];
            QF v = qd[i] + h * qdd[i];
            QF x;
            {
                // velocity limit and inelastic stops as ONE pair of bounds (absent limits are +-inf, mppi_pack.hpp: no branches).
                // joint_limit() of mppi_device.hpp says: at a stop the velocity becomes the displacement that happened,
                // ve = (limit - x_old) / h, never pointing back out of the range.  x_old + h v < lo is the same as v < ve_lo, so
                // the stop is the unconditional bound v >= min(ve_lo, 0) - and the two clamps compose into one because both
                // intervals contain 0: v in [med3(ve_lo, -vmax, 0), med3(ve_hi, 0, vmax)]
                const QF lo = qrep(blk0[i].lower), hi = qrep(blk0[i].upper), z = qrep(0.f);  // (absent: -inf, +inf)
                const QF vlo = qclamp((lo - q[i]) * inv_h, qrep(-b.vmax), z), vhi = qclamp((hi - q[i]) * inv_h, z, qrep(b.vmax));
                v = qclamp(v, vlo, vhi);
                x = qclamp(q[i] + h * v, lo, hi);
            }
            q[i] = x;
            qd[i] = v;
        });
        quad_fk_from<T, JT>(blk0, q, P);
    }
}

// pose of moving body `body` (wave-uniform; < 0: the base) out of the kinematics: a binary tree of UNIFORM branches - three
// scalar compares and four copies.  (Written as arithmetic selects, R += [body == i] R_i, it was a compare, a select and four
// multiply-adds per body: ~70 vector instructions per horizon step for a choice that never changes during a rollout; the
// empty asm keeps the optimiser from turning the branches back into selects.)
template <class T, int JT, int LO, int HI>
MPPI_HD void quad_body_pose(const QPose<T, JT> &P, int body, QM3 &Rb, QF &pb) {
    if constexpr (HI - LO == 1) {
        if constexpr (LO < 0) {
            Rb = P.rot_base();
            pb = P.pos_base();
        } else {
            Rb = P.rot(LO);
            pb = P.pos(LO);
        }
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("");
#endif
    } else {
        constexpr int MID = LO + (HI - LO) / 2;
        if (body < MID) quad_body_pose<T, JT, LO, MID>(P, body, Rb, pb);
        else quad_body_pose<T, JT, MID, HI>(P, body, Rb, pb);
    }
}

// world pose of link l: row r of R (standard) and component r of p
template <class T, class M, int JT>
MPPI_HD void quad_link_pose(M &m, const QPose<T, JT> &P, int l, QM3 &R, QF &p) {
    auto &L = m.l[l];
#if defined(__HIP_DEVICE_COMPILE__)
    const int body = __builtin_amdgcn_readfirstlane(L.body);  // (from the LDS copy of the model it arrives in a vector register)
#else
    const int body = L.body;
#endif
    QM3 Rb;
    QF pb;
    quad_body_pose<T, JT, -1, T::NB>(P, body < 0 ? -1 : (body < T::NB ? body : T::NB - 1), Rb, pb);
    for (int c = 0; c < 3; c++) R.c[c] = Rb.c[0] * L.R[c] + Rb.c[1] * L.R[3 + c] + Rb.c[2] * L.R[6 + c];
    p = pb + Rb.c[0] * L.p[0] + Rb.c[1] * L.p[1] + Rb.c[2] * L.p[2];
}

// Per-rollout constants of the step loop, staged next to the robot model (LDS on the device).  Through the scalar cache
// every one of them was a reload per horizon step - ~11 dependent round trips of a few hundred cycles each that the
// single wavefront of a SIMD cannot hide (lgkmcnt is shared with the LDS traffic, so scalar loads cannot be left in
// flight either).  From LDS they arrive in order a few dozen cycles after the request.
struct StepConsts {
    CtrlBlock u_min, u_max, inv_sigma;
    float goal[4];                       // cost target (actor position, or the fixed goal of POINT_REACH)
    float w[4];                          // cost weights
    CtrlBlock Urow[MPPI_MAX_H];          // nominal control rows, one 64-byte block per horizon step (zero beyond nu)
};
#if defined(__HIP_DEVICE_COMPILE__)
typedef const MPPI_LDS_AS StepConsts LStep;
#else
typedef const StepConsts LStep;
#endif
// entry j of the staging copy (cooperative: lane j of the wavefront fills entries j, j + 64, ...)
MPPI_HD float step_const_entry(CCfg &cfg, CCost &c, const float *root, const float *U, int j) {
    constexpr int kU = (int)(offsetof(StepConsts, Urow) / sizeof(float));
    if (j < 16) return cfg.u_min.v[j];
    if (j < 32) return cfg.u_max.v[j - 16];
    if (j < 48) return cfg.inv_sigma.v[j - 32];
    if (j < 52) {
        const int a = c.actor[0], i = j - 48;
        if (i == 3) return 0.f;
        if (a >= 0) return root[13 * a + i];
        return i < 2 ? c.w[1 + i] : 0.f;  // POINT_REACH without a goal actor: (w1, w2) is the target
    }
    if (j < 56) return c.w[j - 52];
    const int t = (j - kU) >> 4, col = (j - kU) & 15;
    return (t < cfg.H && col < cfg.nu) ? U[t * cfg.nu + col] : 0.f;
}
MPPI_HD int step_const_count(CCfg &cfg) { return (int)(offsetof(StepConsts, Urow) / sizeof(float)) + cfg.H * 16; }

// R, p: world pose of the cost link (PANDA_REACH; computed by the caller, who may share it with the rollout visualisation)
template <class T>
MPPI_HD QF quad_stage_cost(int kind, LStep &sc, const QF *q, const QM3 &R, QF p) {
    if (kind == kCostPointReach) {
        const QF dx = q[0] - sc.goal[0], dy = q[T::NB > 1 ? 1 : 0] - sc.goal[1];
        return sc.w[0] * qsqrt(dx * dx + dy * dy);
    }
    if (kind == kCostPandaReach) {
        const QF d = p - qsel(sc.goal[0], sc.goal[1], sc.goal[2]);
        const QF dist = qsqrt(qsum(d * d));
        // row 2 of R lives in lane 2: R20, R21, R22 -> replicated (see stage_cost in mppi_device.hpp)
        const QF r20 = bc<2>(R.c[0]), r21 = bc<2>(R.c[1]), r22 = bc<2>(R.c[2]);
        const QF a0 = qatan2(r21, -r22);
        const QF a1 = qasin(qclamp(r20, qrep(-1.f), qrep(1.f)));
        return sc.w[0] * dist + sc.w[1] * qsqrt(a0 * a0 + a1 * a1);
    }
    return qrep(0.f);
}

// element `byte_off / 4` of a sample-minor buffer: uniform base pointer + 32-bit BYTE offset = the scalar-base form of the global
// load / store (`global_load_dword v, v_off, s[base]`: one add for the offset) instead of a 64-bit address per element (an add and
// a 64-bit shift-add); mppi_pack.hpp keeps H nu K below 2^30 elements
MPPI_HD float ld32(const float *base, unsigned byte_off) { return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off); }
MPPI_HD void st32(float *base, unsigned byte_off, float v) { *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off) = v; }
// control rows of step t: the nominal row from the staged copy, this sample's noise from HBM (requested one step ahead)
template <int MAXC>
MPPI_HD void load_controls_q(LStep &sc, const float *eps, const float *prior, int nu, int K, int t, int k, ControlRows<MAXC> &r) {
    const CtrlBlock ur = load_block<CtrlBlock>(sc.Urow[t]);  // the whole nominal row in one aligned read
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        const int cc = c < nu ? c : nu - 1;
        r.Ut[c] = ur.v[c];
        r.e[c] = ld32(eps, ((unsigned)(t * nu + cc) * (unsigned)K + (unsigned)k) * 4u);
        r.pr[c] = 0.f;
    }
    if (prior != nullptr) {  // ONE uniform branch for the whole row (rare: use_priors)
#pragma unroll
        for (int c = 0; c < MAXC; c++) r.pr[c] = prior[t * nu + (c < nu ? c : nu - 1)];
    }
}
// `leader` is not consulted: the four lanes of a quad hold the same du and store it to the same address (one dword of
// traffic either way), which replaces seven exec-mask regions per step by plain stores under uniform conditions.
// PLAIN: the common case as a compile-time fact - no null-action / prior sample among the samples of this wavefront, every
// control of the row in use (nu == MAXC), signed control cost: without the four selects per control that the general form pays
// for them on every step (the caller picks with one uniform branch).
template <int MAXC, bool PLAIN>
MPPI_HD float apply_controls_q(LStep &sc, float lambda, bool abs_cost, int nu, int K, const ControlRows<MAXC> &r, int t, int k, bool is_null,
                               bool is_prior, bool /*leader*/, float *du, float *u) {
#pragma unroll
    for (int c = MAXC; c < kMaxNu; c++) u[c] = 0.f;
    const CtrlBlock lo = load_block<CtrlBlock>(sc.u_min), hi = load_block<CtrlBlock>(sc.u_max), is = load_block<CtrlBlock>(sc.inv_sigma);
    float ctrl = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        float v = r.Ut[c] + r.e[c];
        if constexpr (!PLAIN) {
            if (is_null) v = 0.f;
            if (is_prior) v = r.pr[c];
        }
        v = clampf(v, lo.v[c], hi.v[c]);
        const bool on = PLAIN || c < nu;
        u[c] = on ? v : 0.f;
        const float d = v - r.Ut[c];
        if (on) st32(du, ((unsigned)(t * nu + c) * (unsigned)K + (unsigned)k) * 4u, d);
        const float term = r.Ut[c] * d * is.v[c];  // inv_sigma is zero beyond nu
        ctrl += lambda * ((!PLAIN && abs_cost) ? fabsf(term) : term);
    }
    return ctrl;
}

// Whole-horizon rollout of the sample owned by this quad.  Every lane of the quad returns the same S.
// `leader` is true in exactly one lane of the quad (it performs the du store); lanes 0..2 store viz.
// DUMP: the state after every step goes to `traj` as well (q rows [NB][H*K], then qd rows: column t*K + k) - the generic
// Objective mode evaluates Python costs on the materialised trajectory (mppi_rollout_trajectory).
template <class T, int JT, bool DUMP = false, class M = void, class AB = QuadAba>
MPPI_HD QF quad_rollout(M &m0, CCfg &cfg0, CCost &cost0, LStep &sc, const float *dof0, const float *root, const float *eps,
                        const float *prior, float *du, float *viz, int k, bool leader, int row, bool viz_lane, float *traj = nullptr,
                        const AB &ab = AB{}) {
    constexpr int NB = T::NB;
    // read once, kept in SGPRs across the horizon (not laundered)
    const int K = cfg0.K, nu = cfg0.nu, H = cfg0.H, kind = cost0.kind, link = cost0.link[0], viz_link = cfg0.viz_link;
    const float lambda = cfg0.lambda, gamma = cfg0.gamma;
    const bool abs_cost = cfg0.noise_abs_cost != 0, want_viz = cfg0.want_rollouts && viz != nullptr;
    const bool cmd_identity = m0.cmd_identity != 0, need_link = kind == kCostPandaReach;
    const int g = cfg0.k_offset + k;
    const bool is_null = cfg0.sample_null_action && g == cfg0.k_total - 1;
    const bool is_prior = cfg0.use_priors && prior != nullptr && g == cfg0.k_total - 2;
    // the first noise row is requested before anything else: its round trip to HBM overlaps the initial kinematics
    constexpr int MAXC = NB < kMaxNu ? NB : kMaxNu;  // nu <= NB: one command per driven body at most
    ControlRows<MAXC> rows;
    load_controls_q<MAXC>(sc, eps, prior, nu, K, 0, k, rows);
#if defined(__HIP_DEVICE_COMPILE__)
    const bool special_here = __builtin_amdgcn_ballot_w64(is_null || is_prior) != 0;  // wave-uniform
#else
    const bool special_here = is_null || is_prior;
#endif
    const bool plain_controls = !special_here && nu == MAXC && !abs_cost;
    QF q[NB], qd[NB], target[NB];
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        q[i] = qrep(dof0[2 * i]);
        qd[i] = qrep(dof0[2 * i + 1]);
    });
    QF S = qrep(0.f);
    float ctrl = 0.f, disc = 1.f;
    QPose<T, JT> P;  // forward kinematics of the current q, carried across the whole horizon
    quad_base<T>(m0, root, P);
    quad_fk<T>(m0, q, P);
    M *mp = &m0;
    for (int t = 0; t < H; t++) {
        float u[kMaxNu];
        ctrl += plain_controls ? apply_controls_q<MAXC, true>(sc, lambda, abs_cost, nu, K, rows, t, k, is_null, is_prior, leader, du, u)
                               : apply_controls_q<MAXC, false>(sc, lambda, abs_cost, nu, K, rows, t, k, is_null, is_prior, leader, du, u);
        // next step's rows are requested now and consumed after this step's dynamics (the last request re-reads row H-1)
        load_controls_q<MAXC>(sc, eps, prior, nu, K, t + 1 < H ? t + 1 : t, k, rows);
        if (cmd_identity) {  // fixed-base arms, the point robot: one unit-gain command per body
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA { target[ic] = qrep(u[ic < kMaxNu ? (int)ic : 0]); });
        } else {
            M &m = *launder(mp);
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
                constexpr int i = ic;
                const CmdBlock b = load_block<CmdBlock>(m.b[i].cmd);
                float tg = 0.f;
#pragma unroll
                for (int c = 0; c < MAXC; c++) tg += b.v[c] * u[c];
                target[i] = qrep(tg);
            });
        }
        quad_step<T>(*mp, P, q, qd, target, ab);
        if constexpr (DUMP) {  // (the four lanes of a quad hold the same values: same-address stores, as for du)
            const unsigned HK = (unsigned)H * (unsigned)K, col = (unsigned)t * (unsigned)K + (unsigned)k;
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
                constexpr int i = ic;
                traj[(unsigned)i * HK + col] = qlane0(q[i]);
                traj[(unsigned)(NB + i) * HK + col] = qlane0(qd[i]);
            });
        }
        QM3 Rl;  // pose of the cost link, shared with the rollout visualisation when that shows the same link
        QF pl = qrep(0.f);
        for (int c = 0; c < 3; c++) Rl.c[c] = qrep(0.f);
        if (need_link) quad_link_pose<T>(*launder(mp), P, link, Rl, pl);
        S += disc * quad_stage_cost<T>(kind, sc, q, Rl, pl);
        disc *= gamma;
        if (want_viz) {
            QM3 R;
            QF p = pl;
            if (viz_link != link || !need_link) quad_link_pose<T>(*launder(mp), P, viz_link, R, p);
#if defined(__HIP_DEVICE_COMPILE__)
            (void)viz_lane;  // lane 3 mirrors lane 0 (same component, same address): a plain store, no exec-mask region
            viz[((unsigned)(t * 3 + row)) * (unsigned)K + (unsigned)k] = p;  // lane r stores component r
#else
            for (int r = 0; r < 3; r++) viz[((size_t)t * 3 + r) * K + k] = p.v[r];
#endif
        }
    }
    return S + ctrl;
}

}  // namespace mppi
