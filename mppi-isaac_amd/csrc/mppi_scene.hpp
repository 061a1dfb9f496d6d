// mppi_scene.hpp - contact scenes: floating-base robots, free rigid bodies, penalty contact.
//
// Extends the per-sample device code of mppi_device.hpp for scenes in which something can collide
// (reference scenes such as examples/boxer_push: a differential-drive base pushing a block past two
// obstacles on a ground plane).  PhysX's TGS contact solver (reference isaacgym_wrapper.py:30-36) is
// replaced by the build-normative model of DESIGN.md section 3 / SURVEY.md B.5:
//   * primitives: box, sphere, disc (thin wheel cylinder), ground plane z = 0; meshes as AABB boxes;
//   * box-box contacts are vertex-in-box tests in both directions, box/disc/sphere-ground analytic;
//   * contact against STATIC geometry is integrated implicitly: the penalty damper and the stick-regime
//     friction of a contact point are a 6x6 damping matrix that is added to the articulated inertia of
//     the body it acts on (exact for forces linear in that body's own acceleration), so wheel traction
//     and resting contact are unconditionally stable at h = 25 ms;
//   * contact between two DYNAMIC bodies is an explicit spring-damper with Coulomb-capped viscous friction.
// Dynamic frames (robot bodies, floating base, free actors), their force/damping accumulators and the
// net contact force per rigid body are staged in per-lane LDS rows (lane-minor, bank-conflict free) so
// that shapes can address them with run-time indices; the host harness uses a plain array instead.
#pragma once
#include <type_traits>
#include "mppi_device.hpp"
#include <stdio.h>
#include <stdlib.h>

namespace mppi {

// Section clocks (instrumented build only: MPPI_BUILD_VARIANT=sec -> -DMPPI_SECTION_CLOCKS, tools/exp/section_clocks.py):
// MPPI_SEC(id) charges the shader-clock time since the previous mark to section `id`, per wavefront, in a small static LDS
// array that the rollout kernel writes out behind its wave-clock rows.  Compiled out of the product build.
#if defined(MPPI_SECTION_CLOCKS) && defined(__HIP_DEVICE_COMPILE__)
constexpr int kSections = 16;
__device__ __forceinline__ unsigned long long *section_counters() {
    __shared__ unsigned long long s_sec[kSections + 1];
    return s_sec;
}
#define MPPI_SEC(id)                                                   \
    do {                                                               \
        if (threadIdx.x == 0) { /* (kernels with a helper wavefront: the owner's clocks) */ \
            unsigned long long *sc_ = section_counters();              \
            const unsigned long long now_ = __builtin_readcyclecounter(); \
            sc_[id] += now_ - sc_[kSections];                          \
            sc_[kSections] = now_;                                     \
        }                                                              \
    } while (0)
#else
#define MPPI_SEC(id) do { } while (0)
#endif
// Duplication profiling (MPPI_BUILD_VARIANT=dup<k> -> -DMPPI_DUP=k, tools/exp/dup_costs.sh): section k of the pair loop is
// executed TWICE with the second result thrown away (kept alive for the optimiser only), so the physics is bit-identical and
// the kernel-time difference to the product build IS the cost of that section - no timer instructions in the way.
#if defined(MPPI_DUP) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void dup_keep(float x) { asm volatile("" ::"v"(x)); }
__device__ __forceinline__ float launder_zero() { float z = 0.f; asm volatile("" : "+v"(z)); return z; }  // a zero the optimiser cannot see
#endif
// section ids: 0 kinematics + frame stores, 1 accumulator clears + shape poses, 2 dealt broad phase, 3 pair loop,
// 4 inertias / bias forces, 5 first articulated solve, 6 saturation check + second solve, 7 integration + free bodies,
// 8 control sampling + command map, 9 stage cost + rollout point, 10 everything else (init, record tail)

// per-sample indexed scratch ("working set staged in LDS"): element i of this lane lives at p[i*stride]
struct LMem {
    float *p;
    int stride;
    // wave-shared copy of the model's shape records and pair geometry blocks (rollout kernels whose lanes share a sample:
    // the dealt passes index them PER LANE, which through the constant address space means 64-byte vector loads from
    // memory per lane and trip; from LDS it is four ds_read_b128).  Null: read the model.
    const unsigned *tab = nullptr;
    // kSplitOctPair (two wavefronts per sample group): row index of the helper wavefront's own accumulator set
    // ([NF][27] wrench / damping rows, then [n_rb][3] contact-force rows - the layout of set 0 from kAcc on) and of the two
    // words in which it hands its "touched" masks to the first wavefront
    int set1 = 0, xch = 0;
    // kSplitOctPair: row index of the region in which the owner wavefront parks the sample's state while the candidate pairs
    // are walked (state_park; 0: none)
    int park = 0;
    // origin of the rollout's coordinates (root_relative): added back to the positions a rollout writes out
    float ox = 0.f, oy = 0.f;
    // 11 x this sample's position in the row group (the component-minor shape-pose cache, shape_cached below)
    int cm = 0;
    // records of the LIGHT bodies' pairs of this substep ("light bodies" below): element i of this sample at lp[i * lstride] - the tail
    // of the sample's LDS rows in the kernels whose lanes share a sample, a per-lane array in the one-lane kernels; null: none
    float *lp = nullptr;
    int lstride = 0;
    MPPI_HD float &lt(int i) const { return lp[(size_t)i * lstride]; }
    // octet layout of the articulated-body solve (mppi_scene_oct.hpp; rollout kernels of fixed-base trees of more than four bodies):
    // LDS address of THIS lane's view of the model's body blocks (angular lanes: the staged model's own, linear lanes: the copy
    // without inertia tensors); 0: the kernel has none - quad-layout solve
    unsigned oct_bodies = 0;
#if defined(MPPI_CHECK)
    // check build (MPPI_BUILD_VARIANT=check, tests/test_gpu_check_build.py): every access to the sample's rows is bounds-checked
    // against the row length the kernel allocated; a violation traps (the launch fails instead of corrupting a neighbour's row)
    int limit = 0x7fffffff;
    MPPI_HD float &operator[](int i) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if ((unsigned)i >= (unsigned)limit) __builtin_trap();
#endif
        return p[(size_t)i * stride];
    }
#else
    MPPI_HD float &operator[](int i) const { return p[(size_t)i * stride]; }
#endif
};
// Workgroup barriers of the kernels in which two wavefronts hand data over through LDS (owner / helper, mppi_scene.hpp
// kSplitOctPair).  Product build: a plain barrier.  Check build: a PHASE CANARY - every wavefront posts the id of the barrier it
// believes it is at, and after the barrier all posted ids must agree; a wavefront that took another path through the
// barrier sequence (a missed or an extra barrier on one side - the failure mode of such hand-offs) traps at the first
// barrier where the two disagree instead of silently pairing the wrong phases.
#if defined(MPPI_CHECK) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void check_barrier(int id) {
    __shared__ int s_phase[16];
    const int w = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
    if ((threadIdx.x & 63) == 0) s_phase[w] = id;
    __syncthreads();
    for (int j = 1; j < nw; j++)
        if (s_phase[j] != s_phase[0]) __builtin_trap();
    __syncthreads();
}
#define MPPI_BARRIER(id) check_barrier(id)
#elif defined(__HIP_DEVICE_COMPILE__)
#define MPPI_BARRIER(id) __syncthreads()
#else
#define MPPI_BARRIER(id) do { } while (0)
#endif
// Rollouts of a floating-base scene run in coordinates RELATIVE to where the robot starts (x, y; the ground stays z = 0).
// The world-frame spatial algebra refers every inertia, wrench and velocity to the coordinate origin: two metres away from it
// a 270-kg base carries m |c|^2 = 6 times its own yaw inertia as the parallel-axis term, and fp32 loses in the cancellation
// what fp64 does not - at the recorded closed-loop pushing state (robot at (2.1, 1.4)) the fp32 / fp64 gap of the SAME code was a
// median 3e-4 of the cost, 1e-5 with the robot at the origin.  The physics is translation invariant and every cost term is a
// function of differences (constant points of a cost program are shifted by SceneEnv), so the start state is staged with the
// robot's (x, y) subtracted from every actor's row; rollout_scene adds it back to what it writes out (visualisation points,
// dumped trajectories).  out[13 * n_actors]; origin = 0 for fixed-base robots (they stand where their model says).
MPPI_HD float root_relative_entry(const float *root, int j, float ox, float oy) {
    const int c = j % 13;
    return root[j] - (c == 0 ? ox : (c == 1 ? oy : 0.f));
}
template <class M>
MPPI_HD void root_origin(M &m, const float *root, float &ox, float &oy) {
    // (through the constant address space: a scalar load - the two values are wave-uniform and live for the whole rollout)
    const MPPI_CONST_AS float *r = (const MPPI_CONST_AS float *)root;
    ox = m.floating ? r[13 * m.robot_actor] : 0.f;
    oy = m.floating ? r[13 * m.robot_actor + 1] : 0.f;
}
// layout of the table: n_shapes records of kTabShape dwords (DevShape as is), then n_pairs geometry blocks of 16 dwords at a
// pitch of kTabPair = 20: the lanes of an octet read eight CONSECUTIVE blocks with 16-byte accesses (dealt broad phase), and at
// a pitch of 16 dwords blocks j and j + 2 start in the same bank (4-way conflicts on every access); at 20 the eight accesses
// cover the 32 banks once (20 j mod 32 = 0, 20, 8, 28, 16, 4, 24, 12)
constexpr int kTabShape = (int)(sizeof(DevShape) / 4);
constexpr int kTabPair = 20;
static_assert(sizeof(DevShape) == 80 && sizeof(PairGeom) == 64 && sizeof(DevPair) == 128, "table layout");
MPPI_HD constexpr int scene_table_dwords(int n_shapes, int n_pairs) { return kTabShape * n_shapes + kTabPair * n_pairs; }
// cooperative fill by the 64 lanes of the wavefront
template <class M>
MPPI_HD void scene_table_fill(M &m, unsigned *tab, int lane, int lanes) {
    const int ns = m.n_shapes, np = m.n_pairs;
    for (int i = lane; i < kTabShape * ns; i += lanes) tab[i] = reinterpret_cast<const MPPI_CONST_AS unsigned *>(&m.sh[0])[i];
    for (int i = lane; i < 16 * np; i += lanes)
        tab[kTabShape * ns + kTabPair * (i >> 4) + (i & 15)] = reinterpret_cast<const MPPI_CONST_AS unsigned *>(&m.pr[i >> 4].g)[i & 15];
}
struct TabShape {  // the fields shape_world() reads, from the table
    int ent, src_actor;
    float R[9], p[3];
};
MPPI_HD TabShape tab_shape(const LMem &L, int i) {
    const unsigned *r = L.tab + kTabShape * i;
    TabShape S;
    S.ent = (int)r[0];
    S.src_actor = (int)r[1];
    for (int j = 0; j < 9; j++) S.R[j] = __builtin_bit_cast(float, r[8 + j]);
    for (int j = 0; j < 3; j++) S.p[j] = __builtin_bit_cast(float, r[17 + j]);
    return S;
}
template <class M>
MPPI_HD PairGeom tab_pair_geom(M &m, const LMem &L, int ip) {
    PairGeom G;
    __builtin_memcpy(&G, L.tab + kTabShape * m.n_shapes + kTabPair * ip, 64);
    return G;
}

template <class T>
struct SceneLayout {
    static constexpr int NF = T::NB + T::NBASE + kFreeSlots;  // dynamic frames: bodies, bases (one per tree of a moving-base forest), free actors
    static constexpr int kFrame = 0;                 // [NF][18]: R(9) p(3) w(3) vO(3)
    static constexpr int kAcc = NF * 18;             // [NF][27]: f(6) C(21)
    static constexpr int kCf = kAcc + NF * 27;       // [n_rb][3] net contact force
    // then [n_rnd][5]: this sample's size deltas xyz, mass scale, friction of the noisy actors (m.rnd_slot)
    // then (shared-sample kernels only) [n_shapes][12]: world pose R(9) p(3) of every collision shape, refreshed once per
    // substep instead of once per candidate pair
    MPPI_HD static constexpr int floats(int n_rb, int n_rnd, int n_cached_shapes = 0) { return kCf + 3 * n_rb + 5 * n_rnd + 12 * n_cached_shapes; }
};

// counter-based uniform in (0,1): the seeded stand-in for the reference's unseeded np.random draws per env
MPPI_HD float hash_uniform(int seed, int g, int actor, int k) {
    uint32_t h = (uint32_t)seed * 0x9E3779B1u ^ (uint32_t)(g + 1) * 0x85EBCA77u ^ (uint32_t)(actor + 1) * 0xC2B2AE3Du ^ (uint32_t)(k + 1) * 0x27D4EB2Fu;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
MPPI_HD float std_normal_from_uniform(float u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return normcdfinvf(u);
#else
    // Acklam's rational approximation (|error| < 1.2e-9 relative) - host build only
    const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01};
    const double c[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00};
    double p = u, x;
    if (p < 0.02425) { double q = sqrt(-2 * log(p)); x = (((((c[0]*q+c[1])*q+c[2])*q+c[3])*q+c[4])*q+c[5]) / ((((d[0]*q+d[1])*q+d[2])*q+d[3])*q+1); }
    else if (p <= 1 - 0.02425) { double q = p - 0.5, r = q*q; x = (((((a[0]*r+a[1])*r+a[2])*r+a[3])*r+a[4])*r+a[5])*q / (((((b[0]*r+b[1])*r+b[2])*r+b[3])*r+b[4])*r+1); }
    else { double q = sqrt(-2 * log(1 - p)); x = -(((((c[0]*q+c[1])*q+c[2])*q+c[3])*q+c[4])*q+c[5]) / ((((d[0]*q+d[1])*q+d[2])*q+d[3])*q+1); }
    return (float)x;
#endif
}

// this sample's draws for the noisy actors: size deltas, mass scale, friction
template <class T, class M>
MPPI_HD void scene_randomise(M &m, int g, const LMem &L) {
    using Lay = SceneLayout<T>;
    if (m.n_rnd == 0) return;
    for (int a = 0; a < m.n_actors; a++) {
        const int slot = m.rnd_slot[a];
        if (slot < 0) continue;
        const int o = Lay::kCf + 3 * m.n_rb + 5 * slot;
        for (int j = 0; j < 3; j++) {
            const float sg = m.noise[a][j];
            L[o + j] = sg != 0.f ? sg * std_normal_from_uniform(hash_uniform(m.rnd_seed, g, a, j)) : 0.f;
        }
        const float pm = m.noise[a][3], pf = m.noise[a][4], mu0 = m.actor_mu[a];
        L[o + 3] = pm != 0.f ? 1.f + pm * (2.f * hash_uniform(m.rnd_seed, g, a, 3) - 1.f) : 1.f;
        L[o + 4] = pf != 0.f ? mu0 * (1.f + pf * (2.f * hash_uniform(m.rnd_seed, g, a, 4) - 1.f)) : mu0;
    }
}
struct ActorDraw {
    float d[3], ms, mu;
};
template <class T, class M>
MPPI_HD ActorDraw actor_draw(M &m, int a, const LMem &L) {
    const int slot = m.rnd_slot[a];
    if (slot < 0) return ActorDraw{{0.f, 0.f, 0.f}, 1.f, m.actor_mu[a]};
    const int o = SceneLayout<T>::kCf + 3 * m.n_rb + 5 * slot;
    return ActorDraw{{L[o], L[o + 1], L[o + 2]}, L[o + 3], L[o + 4]};
}

// the same by LDS slot (candidate pairs carry the slots of their two actors in PairGeom::rnd); slot < 0: nominal
template <class T, class M>
MPPI_HD ActorDraw actor_draw_slot(M &m, int slot, float mu_nominal, const LMem &L) {
    if (slot < 0) return ActorDraw{{0.f, 0.f, 0.f}, 1.f, mu_nominal};
    const int o = SceneLayout<T>::kCf + 3 * m.n_rb + 5 * slot;
    return ActorDraw{{L[o], L[o + 1], L[o + 2]}, L[o + 3], L[o + 4]};
}

// floats of one sample's LDS rows in the kernels whose lanes share a sample (incl. the shape-pose cache)
template <class T, class M>
MPPI_HD int scene_light_base(M &m) { return SceneLayout<T>::floats(m.n_rb, m.n_rnd, m.n_shapes); }
// the light body's region (see "light bodies"): wrench f'(6) and damping C'(21) of its pairs with robot links about its own centre, the
// reference frame, then kLightSlots records { link = its own joint, (sum of the link's C') S_joint (6) }
constexpr int kLightRow = 27, kLightRef = 27, kLightRec = 28, kLightSlots = 4, kLightSlotFloats = 7, kLightFloats = kLightRec + kLightSlots * kLightSlotFloats;
template <class T, class M>
MPPI_HD int scene_row_floats(M &m) { return SceneLayout<T>::floats(m.n_rb, m.n_rnd, m.n_shapes) + (m.n_light_pairs != 0 ? kLightFloats : 0); }
// the light region of a sample with nothing in it: zero row, no reference frame, no link records.  Written once where the region is set
// up (the kernels, tests/hostemu) and again by contact_forces only after a pass that recorded a pair - a reference frame is set
// then: 32 LDS writes per substep that the samples nowhere near the block do not pay
MPPI_HD void light_region_reset(const LMem &L) {
    for (int j = 0; j < kLightRow; j++) L.lt(j) = 0.f;
    L.lt(kLightRef) = __builtin_bit_cast(float, -1);
    for (int sl = 0; sl < kLightSlots; sl++) L.lt(kLightRec + sl * kLightSlotFloats) = __builtin_bit_cast(float, -1);
}

// extra floats per sample row of the kernels with a helper wavefront (kSplitOctPair): its accumulator set, two mask words and
// the accelerations of the free actors it solves (6 each, at xch + 2)
// ... and the region in which the owner parks the sample's state during the pair walk (q, qd, base row, free actors' rows)
template <class T>
constexpr int scene_park_floats() { return 2 * T::NB + 13 + 13 * kFreeSlots + (T::NB ? T::NB : 1) + 3; }  // + drive targets, + S / ctrl / disc of the rollout
template <class T, class M>
MPPI_HD int scene_pair_floats(M &m) { return SceneLayout<T>::NF * 27 + 3 * m.n_rb + 2 + 6 * kFreeSlots + scene_park_floats<T>(); }

constexpr int scene_floats_max(int nb) { return (nb + 1 + kFreeSlots) * 45 + 3 * (kMaxLinks + kMaxActors); }

// dynamic frame of free actor f / of base r
template <class T>
constexpr int free_frame(int f) { return T::NB + T::NBASE + f; }
template <class T>
struct SceneState {
    float q[T::NB ? T::NB : 1], qd[T::NB ? T::NB : 1];
    float base[13];           // robot root row: pos, quat xyzw, linvel, angvel
    float xbase[T::NBASE > 1 ? T::NBASE - 1 : 1][13];  // ... of the further moving-base robots of the env (one-lane kernels only)
    template <int r> MPPI_HD float *base_row() { if constexpr (r == 0) return base; else return xbase[r - 1]; }
    template <int r> MPPI_HD const float *base_row() const { if constexpr (r == 0) return base; else return xbase[r - 1]; }
    float fr[kFreeSlots][13];   // free actors' root rows
    // accumulator rows (per dynamic frame) and net-contact-force rows (per rigid body, when there are <= 32 of them)
    // that the previous contact pass wrote: only those are cleared by the next pass.  All ones = "clear everything"
    // (fresh LDS: start of a rollout, every launch of the step kernels).
    unsigned acc_dirty = ~0u, cf_dirty = ~0u;
};

MPPI_HD V3 vel_at(const SV &v, V3 p) { return v.l + cross(v.a, p); }

// SPD 6x6 solve A x = b (Cholesky, fully unrolled: everything stays in registers)
MPPI_HD SV solve6(const AI &A, SV b) {
    float a[6][6];
    a[0][0] = A.I.xx; a[0][1] = A.I.xy; a[0][2] = A.I.xz; a[1][1] = A.I.yy; a[1][2] = A.I.yz; a[2][2] = A.I.zz;
    a[1][0] = A.I.xy; a[2][0] = A.I.xz; a[2][1] = A.I.yz;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) { a[r][3 + c] = A.H[3 * r + c]; a[3 + c][r] = A.H[3 * r + c]; }
    a[3][3] = A.M.xx; a[3][4] = A.M.xy; a[3][5] = A.M.xz; a[4][4] = A.M.yy; a[4][5] = A.M.yz; a[5][5] = A.M.zz;
    a[4][3] = A.M.xy; a[5][3] = A.M.xz; a[5][4] = A.M.yz;
    float x[6] = {b.a.x, b.a.y, b.a.z, b.l.x, b.l.y, b.l.z};
#pragma unroll
    for (int j = 0; j < 6; j++) {
        float s = a[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) s -= a[j][k] * a[j][k];
        const float inv = frsqrt(s);
        a[j][j] = inv;  // store 1/L_jj
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            float t = a[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) t -= a[i][k] * a[j][k];
            a[i][j] = t * inv;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        float t = x[i];
#pragma unroll
        for (int k = 0; k < i; k++) t -= a[i][k] * x[k];
        x[i] = t * a[i][i];
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        float t = x[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) t -= a[k][i] * x[k];
        x[i] = t * a[i][i];
    }
    return {{x[0], x[1], x[2]}, {x[3], x[4], x[5]}};
}

// rigid-body inertia about the world origin (world axes) of a body posed at (R, p):
// I_O = R Ic R^T + m(|cw|^2 1 - cw cw^T), h = m cw; plus the bias force v x* (I v)
template <class F>
MPPI_HD void rigid_world(const M3 &R, V3 p, float m, V3 hb, F *Ic, const SV &v, AI &A, SV &pA, V3 &h) {
    h = mul(R, hb) + m * p;
    float T9[9];
    for (int r = 0; r < 3; r++) {
        float r0 = R.a[3 * r], r1 = R.a[3 * r + 1], r2 = R.a[3 * r + 2];
        T9[3 * r + 0] = r0 * Ic[0] + r1 * Ic[1] + r2 * Ic[2];
        T9[3 * r + 1] = r0 * Ic[1] + r1 * Ic[3] + r2 * Ic[4];
        T9[3 * r + 2] = r0 * Ic[2] + r1 * Ic[4] + r2 * Ic[5];
    }
    float invm = m > 0.f ? frcp(m) : 0.f;
    V3 cw = invm * h;
    float hh = dot(h, cw);
    A.I.xx = T9[0] * R.a[0] + T9[1] * R.a[1] + T9[2] * R.a[2] + hh - h.x * cw.x;
    A.I.xy = T9[0] * R.a[3] + T9[1] * R.a[4] + T9[2] * R.a[5] - h.x * cw.y;
    A.I.xz = T9[0] * R.a[6] + T9[1] * R.a[7] + T9[2] * R.a[8] - h.x * cw.z;
    A.I.yy = T9[3] * R.a[3] + T9[4] * R.a[4] + T9[5] * R.a[5] + hh - h.y * cw.y;
    A.I.yz = T9[3] * R.a[6] + T9[4] * R.a[7] + T9[5] * R.a[8] - h.y * cw.z;
    A.I.zz = T9[6] * R.a[6] + T9[7] * R.a[7] + T9[8] * R.a[8] + hh - h.z * cw.z;
    A.H[0] = 0.f;  A.H[1] = -h.z; A.H[2] = h.y;
    A.H[3] = h.z;  A.H[4] = 0.f;  A.H[5] = -h.x;
    A.H[6] = -h.y; A.H[7] = h.x;  A.H[8] = 0.f;
    A.M = {m, 0.f, 0.f, m, 0.f, m};
    V3 n = mul(A.I, v.a) + cross(h, v.l);
    V3 f = m * v.l + cross(v.a, h);
    pA = {cross(v.a, n) + cross(v.l, f), cross(v.a, f)};
}

// ---- per-lane frame / accumulator access ------------------------------------------------------
MPPI_HD void frame_store(const LMem &L, int ent, const M3 &R, V3 p, const SV &v) {
    const int o = ent * 18;
    for (int j = 0; j < 9; j++) L[o + j] = R.a[j];
    L[o + 9] = p.x; L[o + 10] = p.y; L[o + 11] = p.z;
    L[o + 12] = v.a.x; L[o + 13] = v.a.y; L[o + 14] = v.a.z;
    L[o + 15] = v.l.x; L[o + 16] = v.l.y; L[o + 17] = v.l.z;
}
MPPI_HD void frame_load(const LMem &L, int ent, M3 &R, V3 &p, SV &v) {
    const int o = ent * 18;
    for (int j = 0; j < 9; j++) R.a[j] = L[o + j];
    p = {L[o + 9], L[o + 10], L[o + 11]};
    v = {{L[o + 12], L[o + 13], L[o + 14]}, {L[o + 15], L[o + 16], L[o + 17]}};
}
MPPI_HD void acc_add(const LMem &L, int base, int ent, const SV &f, const AI *C) {
    const int o = base + ent * 27;
    L[o + 0] += f.a.x; L[o + 1] += f.a.y; L[o + 2] += f.a.z; L[o + 3] += f.l.x; L[o + 4] += f.l.y; L[o + 5] += f.l.z;
    if (C != nullptr) {
        L[o + 6] += C->I.xx; L[o + 7] += C->I.xy; L[o + 8] += C->I.xz; L[o + 9] += C->I.yy; L[o + 10] += C->I.yz; L[o + 11] += C->I.zz;
        for (int j = 0; j < 9; j++) L[o + 12 + j] += C->H[j];
        L[o + 21] += C->M.xx; L[o + 22] += C->M.xy; L[o + 23] += C->M.xz; L[o + 24] += C->M.yy; L[o + 25] += C->M.yz; L[o + 26] += C->M.zz;
    }
}
#if defined(__HIP_DEVICE_COMPILE__)
// x += v inside the LDS unit (ds_add_f32, no return value): one instruction and no read - wait - add - write round trip
__device__ __forceinline__ void lds_add(float &x, float v) { (void)__hip_atomic_fetch_add(&x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// The same accumulation for the kernels whose lanes share a sample: after the cross-lane sum every lane of the sample holds
// the same totals, so ONE lane (`leader`) adds them, with LDS float adds.  One lane per address and LDS operations of a
// wavefront execute in order: the result is deterministic and is the same IEEE sum as the read-modify-write above.
// Measured by duplication (tools/exp/dup_costs.sh): the read-modify-write version was 21 % of the pushing scene's rollout.
__device__ __forceinline__ void acc_add_shared(const LMem &L, int base, int ent, const SV &f, const AI *C, bool leader) {
    if (!leader) return;
    const int o = base + ent * 27;
    lds_add(L[o + 0], f.a.x); lds_add(L[o + 1], f.a.y); lds_add(L[o + 2], f.a.z);
    lds_add(L[o + 3], f.l.x); lds_add(L[o + 4], f.l.y); lds_add(L[o + 5], f.l.z);
    if (C != nullptr) {
        lds_add(L[o + 6], C->I.xx); lds_add(L[o + 7], C->I.xy); lds_add(L[o + 8], C->I.xz);
        lds_add(L[o + 9], C->I.yy); lds_add(L[o + 10], C->I.yz); lds_add(L[o + 11], C->I.zz);
        for (int j = 0; j < 9; j++) lds_add(L[o + 12 + j], C->H[j]);
        lds_add(L[o + 21], C->M.xx); lds_add(L[o + 22], C->M.xy); lds_add(L[o + 23], C->M.xz);
        lds_add(L[o + 24], C->M.yy); lds_add(L[o + 25], C->M.yz); lds_add(L[o + 26], C->M.zz);
    }
}
#endif
MPPI_HD void acc_load(const LMem &L, int base, int ent, SV &f, AI &C) {
    const int o = base + ent * 27;
    f = {{L[o + 0], L[o + 1], L[o + 2]}, {L[o + 3], L[o + 4], L[o + 5]}};
    C.I = {L[o + 6], L[o + 7], L[o + 8], L[o + 9], L[o + 10], L[o + 11]};
    for (int j = 0; j < 9; j++) C.H[j] = L[o + 12 + j];
    C.M = {L[o + 21], L[o + 22], L[o + 23], L[o + 24], L[o + 25], L[o + 26]};
}

// ---- contact points --------------------------------------------------------------------------
struct PairAcc {
    SV f;    // explicit wrench on shape A's side (about the world origin)
    AI C;    // implicit damping on the dynamic side (modes 1, 2)
    V3 rep;  // reported contact force on A (B receives the opposite)
    float wsum;  // mode 0: sum of the points' ramps (patch normalisation, see pair_normalise)
    bool any;
};
MPPI_HD void pair_zero(PairAcc &a) {
    a.f = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    a.C.I = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 9; j++) a.C.H[j] = 0.f;
    a.C.M = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    a.rep = {0.f, 0.f, 0.f};
    a.wsum = 0.f;
    a.any = false;
}

MPPI_HD void pair_add(PairAcc &a, const PairAcc &b) {
    a.f = a.f + b.f;
    add_to(a.C, b.C);
    a.rep = a.rep + b.rep;
    a.wsum += b.wsum;
    a.any = a.any || b.any;
}

// How the feature points of one pair are dealt over the lanes that share a sample: lane `sub` of `n` takes the
// points sub, sub + n, ...  kSplitNone: one lane per sample.  kSplitQuad: the 4 lanes of a quad hold the same
// sample state (replicated arithmetic) and share the sample's LDS rows; the partial pair sums are combined with
// a DPP butterfly so that all four lanes see bit-identical totals.  kSplitEmulate: host restatement of
// kSplitQuad (the four partial sums are formed one after the other).  kSplitOct: EIGHT lanes per sample - two quads that
// replicate the sample's state arithmetic and the quad-layout robot algebra, and deal the contact work (shape poses, broad
// phase, the 2 x 26 feature points of a box pair) over all eight lanes: K/8 wavefronts (one per SIMD at K = 8192) whose
// divergent narrow phase waits for the busiest of 8 samples instead of 16, each lane testing half the points.
// kSplitOctPair: the octet layout with a HELPER WAVEFRONT per sample group (short trees only: their kernels fit the register
// budget of two resident wavefronts per SIMD).  The candidate-pair loop is 70 % of such a kernel and works out of LDS only -
// frames and shape poses in, wrench / damping rows out - so a second wavefront of the workgroup takes every other pair (and
// half of the shape poses), accumulating into a row set of its own; the first wavefront adds that set to its own after the
// barrier, in a fixed order (deterministic), and goes on alone with the solve.  Two resident wavefronts per SIMD overlap
// where one only waits: measured 1.3x the time for 2x the wavefronts (tools/exp/w2_overlap.sh).
// kSplitOctSolve: kSplitOct with the articulated-body solve itself in the octet layout (mppi_scene_oct.hpp; fixed-base trees)
enum { kSplitNone = 0, kSplitQuad = 1, kSplitEmulate = 2, kSplitOct = 3, kSplitOctPair = 4, kSplitOctSolve = 5 };
constexpr bool split_on_device(int split) { return split == kSplitQuad || split == kSplitOct || split == kSplitOctPair || split == kSplitOctSolve; }
constexpr bool split_octet(int split) { return split == kSplitOct || split == kSplitOctPair || split == kSplitOctSolve; }
struct Split {
    int sub, n;
    int wave = 0;  // kSplitOctPair: 0 = the wavefront that owns the sample state, 1 = its helper
};
#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL>
__device__ __forceinline__ float scene_dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
// sum over the quad, identical in all four lanes: (x0+x1)+(x2+x3) in every lane (fp addition commutes)
__device__ __forceinline__ float quad_allsum(float x) {
    x += scene_dpp<0xB1>(x);  // quad_perm [1,0,3,2]
    x += scene_dpp<0x4E>(x);  // quad_perm [2,3,0,1]
    return x;
}
__device__ __forceinline__ unsigned quad_allor(unsigned x) {
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xf, 0xf, true);
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xf, 0xf, true);
    return x;
}
// sum / or over the lanes that share a sample, identical in all of them.  Octet: the quad totals (already identical within
// each quad) are exchanged with row_ror:8 - the two quads of a sample sit 8 lanes apart inside a 16-lane row (the lane map of
// mppi_oct.hpp, shared with the octet layout of the articulated-body solve): q0 + q1 in one quad, q1 + q0 in the other - the
// same bits.  (Until round 4 the quads of a sample were neighbours and this was row_half_mirror.)
template <int SPLIT>
__device__ __forceinline__ float group_allsum(float x) {
    x = quad_allsum(x);
    if constexpr (split_octet(SPLIT)) x += scene_dpp<0x128>(x);
    return x;
}
template <int SPLIT>
__device__ __forceinline__ unsigned group_allor(unsigned x) {
    x = quad_allor(x);
    if constexpr (split_octet(SPLIT)) x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xf, 0xf, true);
    return x;
}
template <int SPLIT>
__device__ __forceinline__ void group_reduce_explicit(PairAcc &a) {  // mode 0 (both bodies dynamic): wrench and reported force only
    auto S = [](float x) { return group_allsum<SPLIT>(x); };
    a.any = S(a.any ? 1.f : 0.f) > 0.f;
    a.f.a = {S(a.f.a.x), S(a.f.a.y), S(a.f.a.z)};
    a.f.l = {S(a.f.l.x), S(a.f.l.y), S(a.f.l.z)};
    a.rep = {S(a.rep.x), S(a.rep.y), S(a.rep.z)};
    a.wsum = S(a.wsum);
}
template <int SPLIT>
__device__ __forceinline__ void group_reduce(PairAcc &a) {
    auto S = [](float x) { return group_allsum<SPLIT>(x); };
    a.any = S(a.any ? 1.f : 0.f) > 0.f;
    a.f.a = {S(a.f.a.x), S(a.f.a.y), S(a.f.a.z)};
    a.f.l = {S(a.f.l.x), S(a.f.l.y), S(a.f.l.z)};
    a.rep = {S(a.rep.x), S(a.rep.y), S(a.rep.z)};
    a.C.I = {S(a.C.I.xx), S(a.C.I.xy), S(a.C.I.xz), S(a.C.I.yy), S(a.C.I.yz), S(a.C.I.zz)};
    for (int j = 0; j < 9; j++) a.C.H[j] = S(a.C.H[j]);
    a.C.M = {S(a.C.M.xx), S(a.C.M.xy), S(a.C.M.xz), S(a.C.M.yy), S(a.C.M.yz), S(a.C.M.zz)};
}
#endif

// One contact point p (world), unit normal n pointing from B to A, penetration depth > 0.
struct Gains {  // per-pair contact parameters of THIS sample (equal to the packed nominal ones without randomisation)
    int mode;
    float mu, k, cn, ct, kh;
    float inv_d0;  // ramp of the velocity-proportional terms over the first contact_ramp_depth of penetration
    float npts;    // nominal patch size of the pair (mode 0: divisor floor of pair_normalise)
};
// Two dynamic bodies (mode 0, explicit law): a fixed 1/npts share per point lets a face-to-face contact of 18 feature points
// carry 4.5 times the nominal stiffness - beyond the stability limit of an explicit spring-damper at h = 25 ms for the lighter
// body, and the contact chatters (the closed-loop pushing state: perturbations grew 1e4-fold per rollout).  The points of such a
// pair carry the FULL gains (packed with npts = 1) and the pair's summed force is divided by max(npts, sum of the points'
// ramps): the patch never exceeds the nominal stiffness, continuously in the number of points that take part.
MPPI_HD void pair_normalise(const Gains &P, PairAcc &a) {
    const float sc = frcp(fmaxf(P.npts, a.wsum));
    a.f = {sc * a.f.a, sc * a.f.l};
    a.rep = sc * a.rep;
    if (P.mode >= 3) {  // (a light body's pair: the implicit law's damping block as well)
        a.C.I = {sc * a.C.I.xx, sc * a.C.I.xy, sc * a.C.I.xz, sc * a.C.I.yy, sc * a.C.I.yz, sc * a.C.I.zz};
        for (int j = 0; j < 9; j++) a.C.H[j] *= sc;
        a.C.M = {sc * a.C.M.xx, sc * a.C.M.xy, sc * a.C.M.xz, sc * a.C.M.yy, sc * a.C.M.yz, sc * a.C.M.zz};
    }
}
MPPI_HD void contact_point(const Gains &P, V3 p, V3 n, float depth, const SV &vA, const SV &vB, PairAcc &acc) {
    const V3 vr = vel_at(vA, p) - vel_at(vB, p);
    const float vn = dot(vr, n);
    const V3 vt = vr - vn * n;
    const float vtn = fsqrt(dot(vt, vt));
    acc.any = true;
    // Hunt-Crossley-style ramp: damper and implicit spring term grow linearly over the first d0 of penetration, so the
    // normal force (and with it the Coulomb-capped friction) is CONTINUOUS at touch-down; inv_d0 = 0: no ramp
    const float ramp = P.inv_d0 > 0.f ? fminf(1.f, depth * P.inv_d0) : 1.f;
    // the stick cap of the friction viscosity ramps in as well: a grazing contact (f_n -> 0) of a body at rest (|v_t| -> 0)
    // would otherwise get the full stick damper c_t from the ratio of two vanishing numbers
    const float ctr = ramp * P.ct;
    if (P.mode == 0) {  // both dynamic: explicit spring-damper, viscous friction capped by the Coulomb cone
        acc.wsum += ramp;
        const float fn = fmaxf(0.f, P.k * depth - ramp * P.cn * vn);
        const float sc = fminf(ctr, P.mu * fn * frcp(vtn + 1e-9f));
        const V3 f = fn * n - sc * vt;
        acc.f = {acc.f.a + cross(p, f), acc.f.l + f};
        acc.rep = acc.rep + f;
        return;
    }
    // one side static: the spring is evaluated at the END of the step, k (depth - h vn+) - hence the h k
    // term in the implicit normal coefficient (unconditionally stable, no bounce at h = 25 ms); the
    // damper acts on approach and on separation (an approach-only damper toggles with the sign of v_n: resting jitter);
    // friction is implicit too (see below)
    // (modes 3 / 4, a light body held by robot links: damper and end-of-step spring are NOT ramped - the ramp's depth scale is the static
    // sag of a body under its own weight, these contacts carry drive forces at a fraction of it, and a spring that is explicit for the
    // most part throws a 22-gram finger link whose drive has saturated back out of the contact substep after substep; the ramp - over
    // 1 / MPPI_LIGHT_RAMP_DIV of that depth, PairGain::inv_d0 - shapes their stick damper and their patch weights)
    float a = P.mode >= 3 ? P.cn + P.kh : ramp * (P.cn + P.kh);
    {  // never adhesive at the start velocity: a <= k depth / v_n while separating (branch-free; the raw reciprocal is enough)
#if defined(__HIP_DEVICE_COMPILE__)
        const float cap = P.k * depth * __builtin_amdgcn_rcpf(fmaxf(vn, 1e-30f));
#else
        const float cap = P.k * depth / fmaxf(vn, 1e-30f);
#endif
        // (a light body's pair: NO cap.  A contact that relaxes by alpha / (alpha + beta)
        // per substep separates faster than its spring alone would push in every substep after an impact; the capped damper of the
        // robot's gains on a 22-gram finger - h a = tens of kilograms - is honey: the finger keeps its rebound velocity until it has
        // left the contact, its effort drive closes it again at full speed, the grip chatters with a period of four substeps and
        // f_n = 0 drops the block.  Uncapped, the implicit solve balances spring, damper and drive within one substep.)
        a = vn > 0.f && P.mode < 3 ? fminf(a, cap) : a;
    }
    // (a light body's pair: the Coulomb limit is spring and damper WITHOUT the implicit spring term - in a relaxing contact exactly the
    // force that presses the link on, the finger drive's 6 N, and nothing for a block that nothing holds against the link)
    const float fn = fmaxf(0.f, P.k * depth - (P.mode >= 3 ? P.cn : a) * vn);
    // Coulomb friction as an implicit secant viscosity b = min(c_t, mu fn / |v_t|): equals the stick damper
    // at small slip, delivers mu*fn while sliding, and - being implicit - can never reverse the slip
    // velocity (an explicit mu*fn chatters: the yaw inertia seen by a wheel contact is far below the mass)
    const float b = fminf(ctr, P.mu * fn * frcp(vtn + 1e-9f));
    const V3 f = (P.k * depth) * n;
    acc.f = {acc.f.a + cross(p, f), acc.f.l + f};
    if (P.mode >= 3) acc.wsum += ramp;  // (a light body's pair: patch weight)
    // C6 = J^T (b 1 + (a-b) n n^T) J,  J = [-[p]x  1]
    const float ab = a - b;
    const V3 mm = cross(p, n);
    const float pp = dot(p, p);
    acc.C.I.xx += b * (pp - p.x * p.x) + ab * mm.x * mm.x; acc.C.I.xy += -b * p.x * p.y + ab * mm.x * mm.y;
    acc.C.I.xz += -b * p.x * p.z + ab * mm.x * mm.z;       acc.C.I.yy += b * (pp - p.y * p.y) + ab * mm.y * mm.y;
    acc.C.I.yz += -b * p.y * p.z + ab * mm.y * mm.z;       acc.C.I.zz += b * (pp - p.z * p.z) + ab * mm.z * mm.z;
    acc.C.H[0] += ab * mm.x * n.x;            acc.C.H[1] += -b * p.z + ab * mm.x * n.y; acc.C.H[2] += b * p.y + ab * mm.x * n.z;
    acc.C.H[3] += b * p.z + ab * mm.y * n.x;  acc.C.H[4] += ab * mm.y * n.y;            acc.C.H[5] += -b * p.x + ab * mm.y * n.z;
    acc.C.H[6] += -b * p.y + ab * mm.z * n.x; acc.C.H[7] += b * p.x + ab * mm.z * n.y;  acc.C.H[8] += ab * mm.z * n.z;
    acc.C.M.xx += b + ab * n.x * n.x; acc.C.M.xy += ab * n.x * n.y; acc.C.M.xz += ab * n.x * n.z;
    acc.C.M.yy += b + ab * n.y * n.y; acc.C.M.yz += ab * n.y * n.z; acc.C.M.zz += b + ab * n.z * n.z;
    // reported force = penalty force evaluated with the substep's start velocities
    acc.rep = acc.rep + f - (ab * vn) * n - b * vr;
}

// The same contact law against the ground plane z = 0 (unit normal +z, static partner): contact_point with n = (0, 0, 1) and
// vB = 0 written out.  Without fast-math the compiler may not drop the products with the zero components of n (NaN / Inf
// semantics), so the general form costs ~170 instructions per point where ~100 do; only exact-zero terms are dropped, the
// results are the same bits.  Ground pairs are always "dynamic body against static geometry" (implicit).
MPPI_HD void contact_point_ground(const Gains &P, V3 p, float depth, const SV &vA, PairAcc &acc) {
    const V3 vr = vel_at(vA, p);
    const float vn = vr.z;
    const float vtn = fsqrt(vr.x * vr.x + vr.y * vr.y);
    acc.any = true;
    const float ramp = P.inv_d0 > 0.f ? fminf(1.f, depth * P.inv_d0) : 1.f;
    float a = ramp * (P.cn + P.kh);
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const float cap = P.k * depth * __builtin_amdgcn_rcpf(fmaxf(vn, 1e-30f));
#else
        const float cap = P.k * depth / fmaxf(vn, 1e-30f);
#endif
        a = vn > 0.f ? fminf(a, cap) : a;
    }
    const float fn = fmaxf(0.f, P.k * depth - a * vn);
    const float b = fminf(ramp * P.ct, P.mu * fn * frcp(vtn + 1e-9f));
    const float fz = P.k * depth;
    acc.f.a.x += p.y * fz; acc.f.a.y += -(p.x * fz);  // p x (0, 0, fz)
    acc.f.l.z += fz;
    const float ab = a - b;
    const float mx = p.y, my = -p.x;  // p x n
    const float pp = dot(p, p);
    acc.C.I.xx += b * (pp - p.x * p.x) + ab * mx * mx; acc.C.I.xy += -b * p.x * p.y + ab * mx * my;
    acc.C.I.xz += -b * p.x * p.z;                      acc.C.I.yy += b * (pp - p.y * p.y) + ab * my * my;
    acc.C.I.yz += -b * p.y * p.z;                      acc.C.I.zz += b * (pp - p.z * p.z);
    acc.C.H[1] += -b * p.z;  acc.C.H[2] += b * p.y + ab * mx;
    acc.C.H[3] += b * p.z;   acc.C.H[5] += -b * p.x + ab * my;
    acc.C.H[6] += -b * p.y;  acc.C.H[7] += b * p.x;
    acc.C.M.xx += b; acc.C.M.yy += b; acc.C.M.zz += b + ab;
    acc.rep.x += -(b * vr.x); acc.rep.y += -(b * vr.y); acc.rep.z += fz - ab * vn - b * vr.z;
}

struct ShapeW {
    M3 R;
    V3 p;
    SV v;
};
template <class SH>
MPPI_HD ShapeW shape_world(SH &S, const float *root, const LMem &L) {
    M3 Rf;
    V3 pf;
    SV vf = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (S.ent >= 0) {
        frame_load(L, S.ent, Rf, pf, vf);
    } else {
        const float *rs = root + 13 * S.src_actor;
        pf = loadv(rs);
        Rf = quat_to_R(rs + 3);
    }
    ShapeW w;
    w.R = mul(Rf, load3(S.R));
    w.p = pf + mul(Rf, loadv(S.p));
    w.v = vf;
    return w;
}

// World poses of the collision shapes in the sample's LDS rows (kernels whose lanes share a sample): the shapes are
// dealt over the lanes; static shapes are posed once per rollout, the others once per substep.
template <class T, class M>
MPPI_HD int shape_cache_base(M &m) { return SceneLayout<T>::kCf + 3 * m.n_rb + 5 * m.n_rnd; }
// The cache region - rows [base, base + 12 n_shapes) of every sample of the row group - is laid out COMPONENT-MINOR, unlike the
// sample-minor rows around it: the twelve floats of (shape i, sample s) are consecutive, at  region + (i * stride + s) * 12.
// A pose is then three 16-byte LDS accesses instead of twelve 4-byte ones, and the lanes that look up DIFFERENT shapes at the
// same time (dealt shape posing, dealt broad phase) no longer meet in one bank: with sample-minor rows shape i's component c of
// sample s sits in bank (96 i + 8 c + s) mod 32 = (8 c + s) mod 32 whatever i is - an 8-way conflict for an octet whose lanes
// hold eight shapes (the bulk of the 45 k conflict cycles per wavefront the SQ counters show for the gripper scene); here the
// eight samples of a 16-byte access cover banks 12 s .. 12 s + 3 (mod 32), all 32 of them once.
// (L.p points at this sample's column of the sample-minor rows: L.cm = 11 x sample turns that into the slot's address.)
// The same layout for the FRAMES region (18 floats per frame, 8-byte accesses) was built and measured, round 3: 5 % fewer LDS
// instructions in the gripper scene's kernel, -0.8 % kernel time there, +0.7 % on the pushing scene, bank conflicts unchanged
// (22 k per wavefront: they are not the frames') - not kept.
MPPI_HD float *shape_cache_slot(const LMem &L, int base, int i) { return L.p + (size_t)(base + 12 * i) * L.stride + L.cm; }
struct Pose12 {
    float v[12];
};
MPPI_HD Pose12 pose12_load(const float *o) {
    Pose12 q;
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 a = reinterpret_cast<const f4 *>(o)[0], b = reinterpret_cast<const f4 *>(o)[1], c = reinterpret_cast<const f4 *>(o)[2];
    q.v[0] = a.x; q.v[1] = a.y; q.v[2] = a.z; q.v[3] = a.w; q.v[4] = b.x; q.v[5] = b.y; q.v[6] = b.z; q.v[7] = b.w;
    q.v[8] = c.x; q.v[9] = c.y; q.v[10] = c.z; q.v[11] = c.w;
#else
    for (int j = 0; j < 12; j++) q.v[j] = o[j];
#endif
    return q;
}
MPPI_HD void pose12_store(float *o, const Pose12 &q) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef float f4 __attribute__((ext_vector_type(4)));
    reinterpret_cast<f4 *>(o)[0] = f4{q.v[0], q.v[1], q.v[2], q.v[3]};
    reinterpret_cast<f4 *>(o)[1] = f4{q.v[4], q.v[5], q.v[6], q.v[7]};
    reinterpret_cast<f4 *>(o)[2] = f4{q.v[8], q.v[9], q.v[10], q.v[11]};
#else
    for (int j = 0; j < 12; j++) o[j] = q.v[j];
#endif
}
// TAB: the caller knows that the wave-shared table exists (octet rollout kernels) - the per-lane fetch of shape records from
// the model is then not even compiled (its 64-bit per-lane address cost the helper-wavefront kernel a spilled register pair)
template <class T, bool TAB = false, class M>
MPPI_HD void shape_cache_update(M &m, const float *root, const LMem &L, Split sp, bool statics) {
    const int base = shape_cache_base<T>(m);
    for (int i = sp.sub; i < m.n_shapes; i += sp.n) {
        ShapeW w;
        if (TAB || L.tab != nullptr) {
            const TabShape S = tab_shape(L, i);
            if ((S.ent < 0) != statics) continue;
            w = shape_world(S, root, L);
        } else {
            auto &S = m.sh[i];
            if ((S.ent < 0) != statics) continue;
            w = shape_world(S, root, L);
        }
        float *o = shape_cache_slot(L, base, i);
        Pose12 q;
        for (int j = 0; j < 9; j++) q.v[j] = w.R.a[j];
        q.v[9] = w.p.x; q.v[10] = w.p.y; q.v[11] = w.p.z;
        pose12_store(o, q);
    }
}
template <class T, class M>
MPPI_HD ShapeW shape_cached(M &m, int i, const LMem &L) {
    const Pose12 q = pose12_load(shape_cache_slot(L, shape_cache_base<T>(m), i));
    ShapeW w;
    for (int j = 0; j < 9; j++) w.R.a[j] = q.v[j];
    w.p = {q.v[9], q.v[10], q.v[11]};
    w.v = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    return w;
}
MPPI_HD SV frame_velocity(const LMem &L, int ent) {
    if (ent < 0) return SV{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    const int o = ent * 18;
    return SV{{L[o + 12], L[o + 13], L[o + 14]}, {L[o + 15], L[o + 16], L[o + 17]}};
}

// Relative pose of two boxes, computed once per pair: Rrel = R_Y^T R_X (Rrel[3*i+j] = y_i . x_j) and t = R_Y^T (p_X - p_Y)
struct BoxRel {
    float R[9];
    V3 t;
};
MPPI_HD BoxRel box_relative(const ShapeW &X, const ShapeW &Y) {
    BoxRel r;
    const V3 d = X.p - Y.p;
    for (int i = 0; i < 3; i++) {
        const V3 yi = {Y.R.a[i], Y.R.a[3 + i], Y.R.a[6 + i]};  // column i of R_Y
        for (int j = 0; j < 3; j++) r.R[3 * i + j] = yi.x * X.R.a[j] + yi.y * X.R.a[3 + j] + yi.z * X.R.a[6 + j];
        const float ti = yi.x * d.x + yi.y * d.y + yi.z * d.z;
        if (i == 0) r.t.x = ti; else if (i == 1) r.t.y = ti; else r.t.z = ti;
    }
    return r;
}
// Separating-axis test on the six face normals (conservative: never separates two boxes that intersect; the edge-edge
// axes are left out, so some disjoint pairs pass - they then simply find no feature point inside)
MPPI_HD bool boxes_apart(const BoxRel &r, const float *hx, const float *hy, float margin) {
    float a[9];
    for (int j = 0; j < 9; j++) a[j] = fabsf(r.R[j]);
    const float t[3] = {r.t.x, r.t.y, r.t.z};
    bool apart = false;
    for (int i = 0; i < 3; i++)  // face normals of Y
        apart = apart || fabsf(t[i]) > hy[i] + a[3 * i] * hx[0] + a[3 * i + 1] * hx[1] + a[3 * i + 2] * hx[2] + margin;
    for (int j = 0; j < 3; j++)  // face normals of X
        apart = apart || fabsf(t[0] * r.R[j] + t[1] * r.R[3 + j] + t[2] * r.R[6 + j]) > hx[j] + a[j] * hy[0] + a[3 + j] * hy[1] + a[6 + j] * hy[2] + margin;
    return apart;
}

// Penetration depth and push-out direction (box frame) of a point inside a box, continuous everywhere in the interior: with
// the distances dx, dy, dz > 0 to the three nearest faces, depth = (dx^-2 + dy^-2 + dz^-2)^-1/2 - a smooth minimum that
// vanishes on every face, equals the nearest-face distance next to a face and blends near edges and corners - and the normal
// is the unit vector along sum_i max(0, depth / d_i - 1/5)^3 n_i - the direction of its gradient with COMPACT SUPPORT (round 4): a
// face further away than five times the depth has no share in it ((depth / d_i)^3 leaked 3e-5 of the side faces' directions into the
// normal of a block resting 8 mm deep on a 0.5-m table: a creep of 1 um/s; the nearest face always keeps depth / d >= 1/sqrt 3; with the cut at 1/2 the blend
// of a thin box - a finger, 4 mm deep in 10 - turned so fast that one recorded gripper rollout in 8192 split from the oracle WITH weight).  (The nearest-face rule switched the direction of the
// force by 90 degrees where two distances tie; a point leaving through a side face kept its front-face force until the end.)
MPPI_HD void box_interior(float dx, float dy, float dz, V3 y, V3 &nl, float &depth) {
    const float ix = frcp(dx), iy = frcp(dy), iz = frcp(dz);
    const float ds = frsqrt(ix * ix + iy * iy + iz * iz);
    float wx = fmaxf(ds * ix - 0.2f, 0.f), wy = fmaxf(ds * iy - 0.2f, 0.f), wz = fmaxf(ds * iz - 0.2f, 0.f);
    wx = wx * wx * wx; wy = wy * wy * wy; wz = wz * wz * wz;
    const float nn = frsqrt(wx * wx + wy * wy + wz * wz);
    nl = {(y.x > 0.f ? wx : -wx) * nn, (y.y > 0.f ? wy : -wy) * nn, (y.z > 0.f ? wz : -wz) * nn};
    depth = ds;
}

// Feature points of box X (8 corners, 12 edge midpoints, 6 face centres) that lie inside box Y; a pure
// vertex test misses boxes that cross like a plus sign (a tall block against a wide chassis face).
// yc = centre of X in Y's frame, col[j] = column j of R_Y^T R_X scaled by the half extent hx[j];
// sign = +1 when X is shape A (normal from B=Y to A=X)
MPPI_HD void box_points_in_box(const Gains &P, V3 yc, const V3 *col, const ShapeW &Y, const float *hy, float sign, const SV &vA, const SV &vB,
                               Split sp, PairAcc &acc) {
    auto point = [&](int c, V3 &y, float &dx, float &dy, float &dz) MPPI_LAMBDA {
        const int c3 = c / 3, c9 = c / 9;
        const float s0 = (float)(c - 3 * c3 - 1), s1 = (float)(c3 - 3 * c9 - 1), s2 = (float)(c9 - 1);
        y = yc + s0 * col[0] + s1 * col[1] + s2 * col[2];
        dx = hy[0] - fabsf(y.x); dy = hy[1] - fabsf(y.y); dz = hy[2] - fabsf(y.z);
        return c != 13 && dx > 0.f && dy > 0.f && dz > 0.f;  // the centre (13) is not a surface feature
    };
    // phase 1: inside tests only (cheap, branch-free); phase 2: the contact arithmetic for the hits.  Lanes that
    // share a wavefront diverge on WHICH points hit - compacting the hits makes the wavefront pay for the
    // largest hit count of a lane instead of for every point that hits in any lane.
    unsigned hits = 0;
    const int trips = (27 + sp.n - 1) / sp.n;  // the same trip count in every lane: a scalar loop, no exec-mask back edge
    for (int it = 0; it < trips; it++) {
        const int c = sp.sub + it * sp.n;
        V3 y;
        float dx, dy, dz;
        if (point(c, y, dx, dy, dz) && c < 27) hits |= 1u << it;
    }
    while (hits != 0) {
        const int j = __builtin_ctz(hits);
        hits &= hits - 1;
        V3 y;
        float dx, dy, dz;
        point(sp.sub + j * sp.n, y, dx, dy, dz);
        V3 nl;
        float depth;
        box_interior(dx, dy, dz, y, nl, depth);
        const V3 pw = Y.p + mul(Y.R, y);
        const V3 n = sign * mul(Y.R, nl);  // outward normal of Y, oriented from B to A
        contact_point(P, pw, n, depth, vA, vB, acc);
    }
}

// Two DYNAMIC boxes (mode 0, round 5): ONE normal for the whole pair, from the separating-axis test, instead of a push-out
// direction per feature point.  The per-point rule (box_interior: towards the nearest face of the OTHER box) fails where it
// matters for two robots - two equal chassis meeting squarely: every corner and edge midpoint of one lies on a face plane of the
// other (only the face centres are inside: half the nominal stiffness), and as soon as the boxes pitch a little the points of
// the top edge are nearer to the other box's TOP face than to its front: they are pushed up, one chassis climbs the other and
// the two end up inside each other.  Here (oracle: box_pair_sat / corners_along):
//   - 15-axis separating-axis test; an axis that separates: no contact.  depth = the smallest overlap;
//   - n = blend of the six face axes with weights max(0, 2 - o_a / o_min)^3 - a pure face normal unless two overlaps are
//     within a factor two of each other -, oriented from B to A;
//   - every feature point inside the other box is pushed along n, its depth = the distance it has to travel along n to leave
//     that box (ray exit: continuous in the point and in n): box_points_along;
//   - a patch that its points under-sample (a lone corner; two edges that cross: NO feature point inside) is filled up to HALF the
//     nominal stiffness: the share npts / 2 - sum of ramps goes to one more contact of the separating-axis depth at p = the
//     incident box's support point (smoothed over +-0.05 in the direction cosine: a face lying flat gives its centre, a tilted
//     one its deepest corner) clamped onto the reference face: box_pair_fill, called with the pair's TOTAL wsum.  Half, not all of
//     it: the stability bound of the explicit law (alpha + 2 beta < 4) is per BODY, and a block held between two fingers sees two
//     patches - filled to the full nominal stiffness the recorded gripper state lost 0.8 % of its 8192 rollouts to fp32 rounding.
// A = X of `rel` (rel.R[3 i + j] = b_i . a_j, rel.t = centre of A in B's frame).  Every lane of a sample computes the same bits.
MPPI_HD V3 tmul3(const M3 &A, V3 v) {  // A^T v
    return {A.a[0] * v.x + A.a[3] * v.y + A.a[6] * v.z, A.a[1] * v.x + A.a[4] * v.y + A.a[7] * v.z, A.a[2] * v.x + A.a[5] * v.y + A.a[8] * v.z};
}
struct BoxSat {
    V3 n;
    bool hit;
};
// the pair's normal from the six face axes (all that the feature points need: an edge-edge axis that separates the boxes leaves no
// point of one inside the other anyway); the edge axes, the depth and the support point are the fill's business (box_pair_fill),
// which few pairs need
// the six face-axis overlaps of two boxes (o > 0 on every axis: no face axis separates them) and what they are made from
struct BoxOverlaps {
    float aC[9];   // |b_i . a_j|
    float t[3];    // centre of A in B's frame
    float tA[3];   // a_j . (pA - pB)
    float oA[3], oB[3];
    float omin;
    bool apart;
};
MPPI_HD BoxOverlaps box_overlaps(const BoxRel &rel, const float *hA, const float *hB) {
    BoxOverlaps v;
    const float *Cm = rel.R;
    for (int j = 0; j < 9; j++) v.aC[j] = fabsf(Cm[j]);
    v.t[0] = rel.t.x; v.t[1] = rel.t.y; v.t[2] = rel.t.z;
    v.omin = 1e30f;
    v.apart = false;
    for (int i = 0; i < 3; i++) {
        v.tA[i] = v.t[0] * Cm[i] + v.t[1] * Cm[3 + i] + v.t[2] * Cm[6 + i];
        v.oB[i] = hB[i] + v.aC[3 * i] * hA[0] + v.aC[3 * i + 1] * hA[1] + v.aC[3 * i + 2] * hA[2] - fabsf(v.t[i]);
        v.oA[i] = hA[i] + v.aC[i] * hB[0] + v.aC[3 + i] * hB[1] + v.aC[6 + i] * hB[2] - fabsf(v.tA[i]);
        v.apart = v.apart || !(v.oB[i] > 0.f) || !(v.oA[i] > 0.f);
        v.omin = fminf(v.omin, fminf(v.oB[i], v.oA[i]));
    }
    return v;
}
MPPI_HD BoxSat box_pair_sat(const BoxRel &rel, const ShapeW &wa, const float *hA, const ShapeW &wb, const float *hB) {
    BoxSat out;
    out.hit = false;
    out.n = {0.f, 0.f, 0.f};
    const BoxOverlaps v = box_overlaps(rel, hA, hB);
    if (v.apart) return out;
    float nB[3], nA[3];
    const float iomin = frcp(v.omin);
    for (int i = 0; i < 3; i++) {
        // (one reciprocal for the six weights; omin / o_a - 1/2 needed six: 0.9 % of the gripper scene's kernel)
        float w = fmaxf(0.f, 2.f - v.oB[i] * iomin);
        nB[i] = w * w * w * (v.t[i] > 0.f ? 1.f : -1.f);
        w = fmaxf(0.f, 2.f - v.oA[i] * iomin);
        nA[i] = w * w * w * (v.tA[i] > 0.f ? 1.f : -1.f);
    }
    const V3 n = mul(wb.R, V3{nB[0], nB[1], nB[2]}) + mul(wa.R, V3{nA[0], nA[1], nA[2]});
    const float nn2 = dot(n, n);
    if (!(nn2 > 1e-12f)) return out;  // (opposite face normals of equal weight cancel: no direction to push in)
    out.n = frsqrt(nn2) * n;
    out.hit = true;
    return out;
}
MPPI_HD void box_pair_fill(const Gains &P, const BoxRel &rel, const ShapeW &wa, const float *hA, const ShapeW &wb, const float *hB, V3 n, PairAcc &acc) {
    const float deficit = 0.5f * P.npts - acc.wsum;  // (HALF the nominal patch, see above)
    if (!(deficit > 0.f)) return;
    const float *Cm = rel.R;
    const BoxOverlaps v = box_overlaps(rel, hA, hB);
    const float *aC = v.aC, *t = v.t, *tA = v.tA, *oA = v.oA, *oB = v.oB;
    const float omin = v.omin, iomin = frcp(v.omin);
    float odepth = omin;
    bool apart = false;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {  // edge axes b_i x a_j (nearly parallel edges: the face axes cover that direction)
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            const float l2 = 1.f - Cm[3 * i + j] * Cm[3 * i + j];
            const float o = hB[i1] * aC[3 * i2 + j] + hB[i2] * aC[3 * i1 + j] + hA[j1] * aC[3 * i + j2] + hA[j2] * aC[3 * i + j1] -
                            fabsf(t[i2] * Cm[3 * i1 + j] - t[i1] * Cm[3 * i2 + j]);
            const bool used = l2 >= 1e-3f;
            apart = apart || (used && !(o > 0.f));
            odepth = used ? fminf(odepth, o * frsqrt(fmaxf(l2, 1e-3f))) : odepth;
        }
    if (apart) return;
    constexpr float kInvTau = 20.f;
    auto sat1 = [](float x) MPPI_LAMBDA { return fminf(1.f, fmaxf(-1.f, x)); };
    float yB[3] = {0.f, 0.f, 0.f}, xA[3] = {0.f, 0.f, 0.f};
    float WB = 0.f, WA = 0.f;
    for (int i = 0; i < 3; i++) {  // B's face i is the reference, A the incident box: in B's frame
        float w = fmaxf(0.f, 2.f - oB[i] * iomin);
        if (!(w > 0.f)) continue;  // (usually five of the six axes)
        w = w * w * w;
        const float sg = t[i] > 0.f ? 1.f : -1.f;
        float y[3] = {t[0], t[1], t[2]};
        for (int j = 0; j < 3; j++) {
            const float sj = hA[j] * sat1(sg * Cm[3 * i + j] * kInvTau);
            for (int l = 0; l < 3; l++) y[l] -= sj * Cm[3 * l + j];
        }
        for (int l = 0; l < 3; l++) y[l] = fminf(hB[l], fmaxf(-hB[l], y[l]));
        y[i] = sg * (hB[i] - 0.5f * oB[i]);
        for (int l = 0; l < 3; l++) yB[l] += w * y[l];
        WB += w;
    }
    for (int j = 0; j < 3; j++) {  // A's face j is the reference, B the incident box: in A's frame
        float w = fmaxf(0.f, 2.f - oA[j] * iomin);
        if (!(w > 0.f)) continue;
        w = w * w * w;
        const float sg = tA[j] > 0.f ? 1.f : -1.f;
        float x[3] = {-tA[0], -tA[1], -tA[2]};  // centre of B in A's frame
        for (int i = 0; i < 3; i++) {
            const float si = hB[i] * sat1(sg * Cm[3 * i + j] * kInvTau);
            for (int l = 0; l < 3; l++) x[l] += si * Cm[3 * i + l];
        }
        for (int l = 0; l < 3; l++) x[l] = fminf(hA[l], fmaxf(-hA[l], x[l]));
        x[j] = -sg * (hA[j] - 0.5f * oA[j]);
        for (int l = 0; l < 3; l++) xA[l] += w * x[l];
        WA += w;
    }
    const V3 pw = frcp(WA + WB) * (WB * wb.p + mul(wb.R, V3{yB[0], yB[1], yB[2]}) + WA * wa.p + mul(wa.R, V3{xA[0], xA[1], xA[2]}));
    PairAcc one;
    one.f = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    one.rep = {0.f, 0.f, 0.f};
    one.wsum = 0.f;
    one.any = false;
    if (P.mode >= 3) {
        one.C.I = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < 9; j++) one.C.H[j] = 0.f;
        one.C.M = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    }
    contact_point(P, pw, n, odepth, wa.v, wb.v, one);  // (mode 0: touches f, rep, wsum only)
    acc.f = {acc.f.a + deficit * one.f.a, acc.f.l + deficit * one.f.l};
    acc.rep = acc.rep + deficit * one.rep;
    acc.wsum += deficit * one.wsum;
    acc.any = true;
    if (P.mode >= 3) {
        acc.C.I.xx += deficit * one.C.I.xx; acc.C.I.xy += deficit * one.C.I.xy; acc.C.I.xz += deficit * one.C.I.xz;
        acc.C.I.yy += deficit * one.C.I.yy; acc.C.I.yz += deficit * one.C.I.yz; acc.C.I.zz += deficit * one.C.I.zz;
        for (int j = 0; j < 9; j++) acc.C.H[j] += deficit * one.C.H[j];
        acc.C.M.xx += deficit * one.C.M.xx; acc.C.M.xy += deficit * one.C.M.xy; acc.C.M.xz += deficit * one.C.M.xz;
        acc.C.M.yy += deficit * one.C.M.yy; acc.C.M.yz += deficit * one.C.M.yz; acc.C.M.zz += deficit * one.C.M.zz;
    }
}
// box_points_in_box with the pair's normal: n (world, from B to A), nl = the direction in which the points of X leave Y, in Y's
// frame (+-R_Y^T n)
MPPI_HD void box_points_along(const Gains &P, V3 yc, const V3 *col, const ShapeW &Y, const float *hy, V3 nl, V3 n, const SV &vA, const SV &vB,
                              Split sp, PairAcc &acc) {
    const V3 inv = {frcp(fmaxf(fabsf(nl.x), 1e-9f)), frcp(fmaxf(fabsf(nl.y), 1e-9f)), frcp(fmaxf(fabsf(nl.z), 1e-9f))};  // (a face the ray runs parallel to is never its exit)
    auto point = [&](int c, V3 &y) MPPI_LAMBDA {
        const int c3 = c / 3, c9 = c / 9;
        const float s0 = (float)(c - 3 * c3 - 1), s1 = (float)(c3 - 3 * c9 - 1), s2 = (float)(c9 - 1);
        y = yc + s0 * col[0] + s1 * col[1] + s2 * col[2];
        return c != 13 && hy[0] - fabsf(y.x) > 0.f && hy[1] - fabsf(y.y) > 0.f && hy[2] - fabsf(y.z) > 0.f;
    };
    unsigned hits = 0;
    const int trips = (27 + sp.n - 1) / sp.n;
    for (int it = 0; it < trips; it++) {
        const int c = sp.sub + it * sp.n;
        V3 y;
        if (point(c, y) && c < 27) hits |= 1u << it;
    }
    while (hits != 0) {
        const int j = __builtin_ctz(hits);
        hits &= hits - 1;
        V3 y;
        point(sp.sub + j * sp.n, y);
        // ray exit from Y along nl, capped at four times the distance to the NEAREST face: a point that enters through a side face (a
        // finger sliding over the block) starts at depth 0 and gains four times its distance from that face until the exit along n
        // takes over - with the plain ray exit its depth jumped to the pair's penetration the moment it was inside
        const float near = fminf(fminf(hy[0] - fabsf(y.x), hy[1] - fabsf(y.y)), hy[2] - fabsf(y.z));
        const float depth = fminf(fminf(fminf((hy[0] - (nl.x > 0.f ? y.x : -y.x)) * inv.x, (hy[1] - (nl.y > 0.f ? y.y : -y.y)) * inv.y),
                                        (hy[2] - (nl.z > 0.f ? y.z : -y.z)) * inv.z), 4.f * near);
        contact_point(P, Y.p + mul(Y.R, y), n, depth, vA, vB, acc);
    }
}

// sphere (centre ps, radius r) against box Y: closest point of the box to the centre; sign = +1 when the
// sphere is shape A (normal from B = box to A = sphere)
MPPI_HD void sphere_in_box(const Gains &P, V3 ps, float r, const ShapeW &Y, const float *hy, float sign, const SV &vA, const SV &vB, PairAcc &acc) {
    const V3 d = ps - Y.p;
    const V3 y = {Y.R.a[0] * d.x + Y.R.a[3] * d.y + Y.R.a[6] * d.z, Y.R.a[1] * d.x + Y.R.a[4] * d.y + Y.R.a[7] * d.z,
                  Y.R.a[2] * d.x + Y.R.a[5] * d.y + Y.R.a[8] * d.z};
    const V3 cl = {fminf(fmaxf(y.x, -hy[0]), hy[0]), fminf(fmaxf(y.y, -hy[1]), hy[1]), fminf(fmaxf(y.z, -hy[2]), hy[2])};
    const V3 e = y - cl;
    const float dist2 = dot(e, e);
    if (dist2 >= r * r) return;
    V3 nl;
    float depth;
    if (dist2 > 1e-12f) {  // centre outside the box: normal along the shortest connection
        const float dist = fsqrt(dist2);
        nl = frcp(dist) * e;
        depth = r - dist;
    } else {  // centre inside: push out through the nearest face
        const float dx = hy[0] - fabsf(y.x), dy = hy[1] - fabsf(y.y), dz = hy[2] - fabsf(y.z);
        box_interior(fmaxf(dx, 1e-9f), fmaxf(dy, 1e-9f), fmaxf(dz, 1e-9f), y, nl, depth);  // (centre on the surface: positive distances)
        depth += r;
    }
    const V3 pw = Y.p + mul(Y.R, cl);
    contact_point(P, pw, sign * mul(Y.R, nl), depth, vA, vB, acc);
}

// disc (thin wheel / caster cylinder: centre pc, unit axis ax, radius r) against box Y - round 5: the wheels and casters of a
// mobile base meet the boxes of OTHER actors too (reference: everything in an env shares one collision group,
// isaacgym_wrapper.py:436-442), not just the ground.  ONE analytic point, like the disc's rim point against the ground - the
// deepest point of the disc in the box: with e the direction from the disc centre INTO the box (towards the box's closest point
// while the centre is outside; against the push-out normal of box_interior once it is inside - the two agree on the surface) and
// e_p its part in the disc's plane, the point is  centre + r e_p / |e|:  the rim point when the box lies in the disc's plane, the
// centre when it lies over the flat side, everything in between continuously; a box thinner than the disc reaches along that
// ray is met in the middle of its stretch [t_in, t_out] (slab test) instead.  Depth and normal are those of a point inside a
// box (box_interior).  sign = +1 when the disc is shape A.
MPPI_HD void disc_in_box(const Gains &P, V3 pc, V3 ax, float r, const ShapeW &Y, const float *hy, float sign, const SV &vA, const SV &vB, PairAcc &acc) {
    const V3 d0 = pc - Y.p;
    const V3 yc = {Y.R.a[0] * d0.x + Y.R.a[3] * d0.y + Y.R.a[6] * d0.z, Y.R.a[1] * d0.x + Y.R.a[4] * d0.y + Y.R.a[7] * d0.z,
                   Y.R.a[2] * d0.x + Y.R.a[5] * d0.y + Y.R.a[8] * d0.z};           // disc centre in the box frame
    const V3 cl = {fminf(fmaxf(yc.x, -hy[0]), hy[0]), fminf(fmaxf(yc.y, -hy[1]), hy[1]), fminf(fmaxf(yc.z, -hy[2]), hy[2])};
    const V3 al = {Y.R.a[0] * ax.x + Y.R.a[3] * ax.y + Y.R.a[6] * ax.z, Y.R.a[1] * ax.x + Y.R.a[4] * ax.y + Y.R.a[7] * ax.z,
                   Y.R.a[2] * ax.x + Y.R.a[5] * ax.y + Y.R.a[8] * ax.z};           // disc axis in the box frame
    V3 e = cl - yc;                                  // into the box: towards its closest point ...
    const float cx = hy[0] - fabsf(yc.x), cy = hy[1] - fabsf(yc.y), cz = hy[2] - fabsf(yc.z);
    if (cx > 0.f && cy > 0.f && cz > 0.f) {          // ... or, with the centre inside, against the push-out direction
        V3 nc;
        float dc;
        box_interior(cx, cy, cz, yc, nc, dc);
        e = -1.f * nc;
    }
    const float le2 = dot(e, e);
    if (!(le2 > 1e-20f)) return;                     // (centre exactly on the surface: no direction - and no depth)
    const V3 ep = e - dot(e, al) * al;               // its part in the disc's plane
    const float ile = frsqrt(le2), lp2 = dot(ep, ep);
    float t = r * fsqrt(lp2) * ile;                  // distance of the contact point from the centre: r sin(angle between e and the axis)
    V3 y = yc;
    if (lp2 > 1e-12f * le2) {
        const V3 u = frsqrt(lp2) * ep;               // ray direction; the box's stretch of the ray by the slab test
        float t_in = -1e30f, t_out = 1e30f;
        const float uc[3] = {u.x, u.y, u.z}, oc[3] = {yc.x, yc.y, yc.z};
        for (int j = 0; j < 3; j++)
            if (fabsf(uc[j]) > 1e-6f) {
                const float inv = frcp(uc[j]), t1 = (-hy[j] - oc[j]) * inv, t2 = (hy[j] - oc[j]) * inv;
                t_in = fmaxf(t_in, fminf(t1, t2));
                t_out = fminf(t_out, fmaxf(t1, t2));
            }
        if (t_out < 1e29f && t_in > -1e29f) t = fminf(t, 0.5f * (fmaxf(t_in, 0.f) + t_out));
        y = yc + t * u;
    }
    const float dx = hy[0] - fabsf(y.x), dy = hy[1] - fabsf(y.y), dz = hy[2] - fabsf(y.z);
    if (!(dx > 0.f && dy > 0.f && dz > 0.f)) return;
    V3 nl;
    float depth;
    box_interior(dx, dy, dz, y, nl, depth);
    contact_point(P, Y.p + mul(Y.R, y), sign * mul(Y.R, nl), depth, vA, vB, acc);
}

// disc (centre pc, unit axis ax, radius r) against a sphere (centre ps, radius rs): the disc's point nearest to the sphere centre -
// the centre's projection into the disc's plane, pulled back onto the disc - against the sphere's surface; sign = +1 when the disc
// is shape A (the normal points from B to A).  (Round 5: wheels of a mobile base against the sphere obstacles of the benchmark
// adapters, reference benchmarks/point_robot/mppi_planner/mppi_planner_wrapper.py:58-79.)
MPPI_HD void disc_sphere(const Gains &P, V3 pc, V3 ax, float r, V3 ps, float rs, float sign, const SV &vA, const SV &vB, PairAcc &acc) {
    V3 e = ps - pc;
    e = e - dot(e, ax) * ax;                               // the sphere centre's offset within the disc's plane
    const float l2 = dot(e, e);
    const V3 q = pc + (l2 > r * r ? r * frsqrt(l2) : 1.f) * e;   // the disc's nearest point
    const V3 d = q - ps;
    const float d2 = dot(d, d);
    if (d2 >= rs * rs) return;
    V3 n = ax;                                             // (the sphere centre ON the disc: push along the axis)
    float dist = 0.f;
    if (d2 > 1e-12f) {
        dist = fsqrt(d2);
        n = frcp(dist) * d;                                // from the sphere towards the disc
    }
    contact_point(P, q, sign * n, rs - dist, vA, vB, acc);
}

// two spheres: normal along the line of centres (from B to A), contact point in the middle of the overlap; coincident centres
// push apart along +z (sphere obstacles of the plannerbenchmark adapters against sphere-shaped links,
// reference benchmarks/panda_arm/mppi_planner/mppi_planner_wrapper.py:58-79)
MPPI_HD void sphere_sphere(const Gains &P, V3 pa, float ra, V3 pb, float rb, const SV &vA, const SV &vB, PairAcc &acc) {
    const V3 d = pa - pb;
    const float dist2 = dot(d, d), rs = ra + rb;
    if (dist2 >= rs * rs) return;
    V3 n = {0.f, 0.f, 1.f};
    float dist = 0.f;
    if (dist2 > 1e-12f) {
        dist = fsqrt(dist2);
        n = frcp(dist) * d;
    }
    const float depth = rs - dist;
    contact_point(P, pb + (rb - 0.5f * depth) * n, n, depth, vA, vB, acc);
}

constexpr int kDealtBroadPhaseMin = 16;  // candidate pairs above which the quad kernels deal the broad phase over the lanes
// larger trees only: on the pushing scene (2-body tree, 11 pairs that are mostly near each other) the dealt pass is pure
// overhead - measured +8 % on the octet kernel at equal state (1.386 -> 1.494 ms), as it was on the quad kernel
// ---- light bodies ------------------------------------------------------------------------------
// A free actor of at most MPPI_LIGHT_BODY_MASS that the robot outweighs MPPI_LIGHT_BODY_RATIO times (the 1-gram block of the reference's
// examples/panda_pick between the fingers of a 17-kg arm; PhysX's implicit solver holds and lifts it, isaacgym_wrapper.py:29-36).  The
// explicit law of two dynamic bodies is as stiff as the LIGHTER body can carry in an explicit step - 1.3 N/m for one gram at
// h = 25 ms: a finger drive closes the fingers THROUGH the block - and its stick damper creeps at g h.  A pair of the light body with
// a robot link (PairGeom::mode 3: A is the link, 4: B; mppi_pack.hpp: one light body per scene, none that touches another free actor)
// takes the implicit law of a static partner on BOTH bodies with the ROBOT's gains, damper and end-of-step spring not ramped
// (contact_point), staggered (oracle: light_pair_t).  With C = J^T (b 1 + (a - b) n n^T) J over the pair's points, f its spring wrench:
//   robot link X:  wrench = -+f - C (v_X+ - v_L)    C joins the link's articulated inertia like a static contact's; the light body is a
//                                                   wall that moves with its velocity at the START of the substep;
//   light body L:  wrench = +-f - C (v_L+ - v_X+)   solved AFTER the robot (free_body_accel): +-f + C v_X - C v_L+ from the contact pass,
//                                                   and the links' velocity CHANGES over the substep as
//                                                       (sum_X C_X) dv_ref + sum_X (C_X S_X) dqd_X
//                                                   ref = the frame nearest the base among the PARENTS of the links in contact, S_X / dqd_X
//                                                   the link's own joint axis / rate change (light_reference_change, step_free_bodies): exact when the links
//                                                   hang off one parent - two fingers on a hand: their relative motion is what a pinch is
//                                                   made of -, one substep late only for joints between `ref` and a link's parent.
// Both solves are unconditionally stable (a squeeze between two fingers contracts by m / (m + h C) per substep); what the robot feels of
// the light body is its weight and an added mass h C while it accelerates.  Its contacts with STATIC geometry keep the law of modes
// 1 / 2 with the gains of its own mass (at rest it lies still to 1e-7 m/s in that law's sag; with the robot's gains the nine feature
// points of a face toggle in and out of a half-micrometre contact for ever).
// fp32: the pair is evaluated in coordinates about the light body's centre `lo` - its 6x6 (inertia 1e-7 kg m^2 next to h C ~ 10 kg about
// the world origin) would lose the body's own inertia to rounding otherwise; the robot's share is shifted to the world origin
// (light_shift), the light body's solve stays about lo.  Storage: kLightFloats per sample - (f', C') about lo, the reference frame,
// kLightSlots records (link, C S) - behind the sample's LDS rows (a per-lane array in the one-lane kernels): 56 floats, the gripper
// scene's kernel keeps four workgroups per CU (a record of C per pair, 112 floats, took it to three: 1.87 -> 3.0 ms).
// Tried and dropped (tests/test_scene_kat.py::test_light_body_law_survives_random_gripper_motion caught both): every link's change
// through an effective contact POINT - a force where the patch has a damping matrix: what falls into a direction the patch damps weakly
// spins a gram to 1000 rad/s -; all changes through one common reference link - two fingers both a substep late: the pinch rings at the
// substep rate for ever.
// 6x6 about the world origin from the one about the point o (motion E = [[1, 0], [-[o]x, 1]]: C_O = E^T C' E); o -> -o: the way back
MPPI_HD AI light_shift(const AI &c, V3 o) {
    // G = [o]x H'^T (3x3), PM = [o]x M'
    const float Ht[9] = {c.H[0], c.H[3], c.H[6], c.H[1], c.H[4], c.H[7], c.H[2], c.H[5], c.H[8]};
    const float Mm[9] = {c.M.xx, c.M.xy, c.M.xz, c.M.xy, c.M.yy, c.M.yz, c.M.xz, c.M.yz, c.M.zz};
    float G[9], PM[9];
    for (int j = 0; j < 3; j++) {  // columns
        const V3 g = cross(o, V3{Ht[j], Ht[3 + j], Ht[6 + j]}), q = cross(o, V3{Mm[j], Mm[3 + j], Mm[6 + j]});
        G[j] = g.x; G[3 + j] = g.y; G[6 + j] = g.z;
        PM[j] = q.x; PM[3 + j] = q.y; PM[6 + j] = q.z;
    }
    // -P M' P = (P M') P^T: row r of PM crossed with o  ((X P^T)_r = -(X_r x o) ... X [o]x^T v = X (v x o)): rows: PMP_r = o x PM_r^T taken column-wise
    float Q[9];  // Q = PM [o]x^T  ->  Q[r][:] = -(PM[r][:] x o) = o x PM[r][:]
    for (int r = 0; r < 3; r++) {
        const V3 q = cross(o, V3{PM[3 * r], PM[3 * r + 1], PM[3 * r + 2]});
        Q[3 * r] = q.x; Q[3 * r + 1] = q.y; Q[3 * r + 2] = q.z;
    }
    AI out;
    out.I = {c.I.xx + 2.f * G[0] + Q[0], c.I.xy + G[1] + G[3] + Q[1], c.I.xz + G[2] + G[6] + Q[2],
             c.I.yy + 2.f * G[4] + Q[4], c.I.yz + G[5] + G[7] + Q[5], c.I.zz + 2.f * G[8] + Q[8]};
    for (int j = 0; j < 9; j++) out.H[j] = c.H[j] + PM[j];
    out.M = c.M;
    return out;
}
MPPI_HD SV light_shift(const SV &f, V3 o) { return SV{f.a + cross(o, f.l), f.l}; }          // wrench about o -> about the world origin
MPPI_HD SV light_motion_at(const SV &v, V3 o) { return SV{v.a, v.l + cross(v.a, o)}; }      // motion about the world origin -> about o
// how far the spatial velocity of the light bodies' REFERENCE frame changed over the substep: the changes dqd of the joint rates - after
// the velocity and joint limits - on the joint axes of the substep's poses, summed down the tree (a floating base: its integrated root
// row minus the velocity its frame row holds); only the reference frame's change is kept (the links' own joints enter the light body's
// solve through dqd itself).  (Until the end of round 6 every frame's change went into the frames' velocity rows and free_body_accel
// read them back: 162 LDS operations per substep + 18 per record, 16 % of the gripper kernel's time with the block in the gripper.)
template <class T, class M>
MPPI_HD SV light_reference_change(M &m, const SceneState<T> &s, const LMem &L, const float *dqd, int ref) {
    constexpr int NB = T::NB;
    SV out = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    SV vb[T::NBASE];
    static_for<0, T::NBASE>([&](auto rc) MPPI_LAMBDA {
        constexpr int r = rc;
        vb[r] = SV{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        if (m.floating) {
            const float *bs = s.template base_row<r>();
            const V3 w = loadv(bs + 10), vl = loadv(bs + 7);
            const SV now = SV{w, vl - cross(w, loadv(bs))};
            const SV was = frame_velocity(L, NB + r);
            vb[r] = SV{now.a - was.a, now.l - was.l};
        }
        if (ref == NB + r) out = vb[r];
    });
    SV v[NB ? NB : 1];
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const int o = i * 18;
        const V3 az = {L[o + 2], L[o + 5], L[o + 8]}, p = {L[o + 9], L[o + 10], L[o + 11]};
        const float dq = dqd[i];
        const SV sj = m.b[i].k0.jtype == 0 ? SV{dq * az, dq * cross(p, az)} : SV{{0.f, 0.f, 0.f}, dq * az};
        if constexpr (par < 0) v[i] = vb[base_of_parent(par)] + sj;
        else v[i] = v[par < 0 ? 0 : par] + sj;
        if (ref == i) out = v[i];
    });
    return out;
}

template <class T>
constexpr bool dealt_broad_phase(int split) { return split_on_device(split) && T::NB > 4; }
// Poses, per-sample sizes and broad-phase verdict of one candidate pair.
struct PairPose {
    ShapeW wa, wb;
    float hA[3], hB[3];
    ActorDraw da, db;
    bool robotA, robotB;
    BoxRel rel;  // (box-box pairs only)
};
// Broad phase: six-axis SAT for two boxes, otherwise the bounding sphere of one shape against the other shape's
// box grown by that radius (both ways).  Conservative by construction (a margin covers rounding), so skipping
// changes no result; it removes the 2 x 26 feature-point tests of the many link-vs-table / link-vs-block
// pairs that are nowhere near each other.  Returns true when the two shapes are certainly apart.
template <class T, bool kCached, class M>
MPPI_HD bool pair_broad_phase(M &m, int ip, const PairGeom &G, const float *root, const LMem &L, PairPose &o) {
    const bool has_b = G.b >= 0;
    if constexpr (kCached) {
        o.wa = shape_cached<T>(m, G.a, L);
        if (has_b) o.wb = shape_cached<T>(m, G.b, L);
    } else {
        o.wa = shape_world(m.sh[G.a], root, L);
        if (has_b) o.wb = shape_world(m.sh[G.b], root, L);
    }
    const ShapeW &wa = o.wa, &wb = o.wb;
    float *hA = o.hA, *hB = o.hB;
    for (int j = 0; j < 3; j++) { hA[j] = G.hA[j]; hB[j] = G.hB[j]; }
    o.da = {{0.f, 0.f, 0.f}, 1.f, 0.f};
    o.db = {{0.f, 0.f, 0.f}, 1.f, 0.f};
    o.robotA = true;
    o.robotB = true;
    if (G.rnd) {  // this sample's own size of the noisy actors in the pair (friction and mass scale: see contact_forces)
        const int sa = ((G.rnd >> 8) & 15) - 1, sb = ((G.rnd >> 12) & 15) - 1;
        o.robotA = sa < 0;
        o.robotB = sb < 0;
        if (sa >= 0) {
            o.da = actor_draw_slot<T>(m, sa, 0.f, L);
            if (G.typeA == 0) for (int j = 0; j < 3; j++) hA[j] += 0.5f * o.da.d[j];
            else if (G.typeA == 1) hA[0] += o.da.d[0];
        }
        if (sb >= 0) {
            o.db = actor_draw_slot<T>(m, sb, 0.f, L);
            if (G.typeB == 0) for (int j = 0; j < 3; j++) hB[j] += 0.5f * o.db.d[j];
            else if (G.typeB == 1) hB[0] += o.db.d[0];
        }
    }
    const int typeA = G.typeA, typeB = G.typeB;
    bool apart = false;
    if (has_b && typeA == 0 && typeB == 0) {
        o.rel = box_relative(wa, wb);
        apart = boxes_apart(o.rel, hA, hB, 1e-4f);
    } else {
        constexpr float kMargin = 1e-4f;
        const float rA = typeA == 0 ? fsqrt(hA[0] * hA[0] + hA[1] * hA[1] + hA[2] * hA[2]) * 1.000001f : hA[0];  // (rounded up: conservative)
        if (!has_b) {
            // ground: a box by its support function along z (lowest corner at p.z - sum |R_zj| h_j: the chassis of a wheeled
            // base hovers within its bounding sphere's reach of the ground for ever), a sphere by its radius
            apart = typeA == 0 ? wa.p.z - (fabsf(wa.R.a[6]) * hA[0] + fabsf(wa.R.a[7]) * hA[1] + fabsf(wa.R.a[8]) * hA[2]) > kMargin
                               : (typeA != 2 && wa.p.z > rA + kMargin);
        } else {   // (a disc against another shape: culled like a sphere of its radius)
            const float rB = typeB == 0 ? fsqrt(hB[0] * hB[0] + hB[1] * hB[1] + hB[2] * hB[2]) * 1.000001f : hB[0];
            const V3 d = wa.p - wb.p;
            if (typeB == 0) {
                const V3 y = {wb.R.a[0] * d.x + wb.R.a[3] * d.y + wb.R.a[6] * d.z, wb.R.a[1] * d.x + wb.R.a[4] * d.y + wb.R.a[7] * d.z,
                              wb.R.a[2] * d.x + wb.R.a[5] * d.y + wb.R.a[8] * d.z};
                apart = fabsf(y.x) > hB[0] + rA + kMargin || fabsf(y.y) > hB[1] + rA + kMargin || fabsf(y.z) > hB[2] + rA + kMargin;
            } else {
                apart = dot(d, d) > (rA + rB + kMargin) * (rA + rB + kMargin);
            }
            if (typeA == 0) {
                const V3 x = {wa.R.a[0] * d.x + wa.R.a[3] * d.y + wa.R.a[6] * d.z, wa.R.a[1] * d.x + wa.R.a[4] * d.y + wa.R.a[7] * d.z,
                              wa.R.a[2] * d.x + wa.R.a[5] * d.y + wa.R.a[8] * d.z};
                apart = apart || fabsf(x.x) > hA[0] + rB + kMargin || fabsf(x.y) > hA[1] + rB + kMargin || fabsf(x.z) > hA[2] + rB + kMargin;
            }
        }
    }
    return apart;
}

// All candidate pairs of the scene -> per-frame wrench / damping accumulators and net contact forces.
template <class T, int SPLIT = kSplitNone, class M = CModel>
MPPI_HD unsigned contact_forces(M &m, const float *root, const LMem &L, unsigned &acc_dirty, unsigned &cf_dirty, Split split = Split{0, 1}) {
    unsigned touched = 0;  // entities (dynamic frames) whose accumulator rows are non-zero after this call
    using Lay = SceneLayout<T>;
    // contacts are sparse: clear only what the previous pass wrote (375 LDS rows per substep in the gripper scene otherwise)
    // accumulator rows this wavefront writes (kSplitOctPair: the helper has its own set, merged below)
    constexpr bool kPair = SPLIT == kSplitOctPair;
    const bool helper = kPair && split.wave != 0;
    const int kAccW = helper ? L.set1 : (int)Lay::kAcc, kCfW = kAccW + (Lay::kCf - Lay::kAcc);
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (kPair) MPPI_BARRIER(1);  // the frames of this substep are written; the previous merge is done
#endif
    for (int e = 0; e < Lay::NF; e++)
        if ((acc_dirty >> e) & 1u)
            for (int j = 0; j < 27; j++) L[kAccW + 27 * e + j] = 0.f;
    const bool cf_all = m.n_rb > 32;
    for (int r = 0; r < m.n_rb; r++)
        if (cf_all || ((cf_dirty >> r) & 1u)) {
            L[kCfW + 3 * r] = 0.f; L[kCfW + 3 * r + 1] = 0.f; L[kCfW + 3 * r + 2] = 0.f;
        }
    unsigned cf_touched = 0;
    const SV zero = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    constexpr bool kCached = SPLIT != kSplitNone;
    // records of the light bodies' pairs ("light bodies" above; never in the kernel with a helper wavefront: mppi_create gives a scene
    // with such pairs the one-wavefront kernels): all free at the start of the pass (light_region_reset: only where the last pass left something)
    int n_light = 0;
    if constexpr (!kPair)
        if (m.n_light_pairs != 0 && (!split_on_device(SPLIT) || split.sub == 0) && __builtin_bit_cast(int, L.lt(kLightRef)) >= 0) light_region_reset(L);
    if constexpr (kPair) {
        // dealt over both wavefronts (the owner posing all shapes before the first barrier, one barrier less, measured the same)
        shape_cache_update<T, true>(m, root, L, Split{split.sub + split.n * split.wave, 2 * split.n}, false);
        MPPI_BARRIER(2);
    } else if constexpr (kCached) {
        shape_cache_update<T, split_octet(SPLIT)>(m, root, L, SPLIT == kSplitEmulate ? Split{0, 1} : split, false);
    }
    MPPI_SEC(1);
    // Dealt broad phase (quad kernels of the larger trees, whose scenes carry a candidate pair per link and obstacle):
    // lane r of the quad tests the pairs r, r + 4, ... on its own and the verdicts are OR-ed over the quad; the pair
    // loop then visits only the survivors.  pair_broad_phase() restates the test of the loop body term by term - the
    // verdicts must agree bit for bit (tests/test_gpu_parity.py compares contact scenes with the oracle either way).
    // Pays when most pairs are apart in every sample of a wavefront: 23-pair gripper scene -17 % away from contact.
    // (one verdict bit per candidate pair: 128 pairs = four words; the reference's ten obstacle spheres around a ten-link arm are 100)
    unsigned alive_lo = ~0u, alive_hi = ~0u, alive_2 = ~0u, alive_3 = ~0u;
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (dealt_broad_phase<T>(SPLIT)) {
        if (m.n_pairs > kDealtBroadPhaseMin) {
            alive_lo = alive_hi = alive_2 = alive_3 = 0u;
            const int trips = (m.n_pairs + split.n - 1) / split.n;
            for (int it = 0; it < trips; it++) {
                const int ip = it * split.n + split.sub;
                const int ic = ip < m.n_pairs ? ip : m.n_pairs - 1;
                const PairGeom G = L.tab != nullptr ? tab_pair_geom(m, L, ic) : load_block<PairGeom>(m.pr[ic].g);
                PairPose pp;
                const bool apart = pair_broad_phase<T, true>(m, ic, G, root, L, pp);
                const unsigned bit = (!apart && ip < m.n_pairs) ? 1u << (ip & 31) : 0u;
                const int w = ip >> 5;
                alive_lo |= w == 0 ? bit : 0u;
                alive_hi |= w == 1 ? bit : 0u;
                alive_2 |= w == 2 ? bit : 0u;
                alive_3 |= w == 3 ? bit : 0u;
            }
            alive_lo = group_allor<SPLIT>(alive_lo);
            alive_hi = group_allor<SPLIT>(alive_hi);
            if (m.n_pairs > 64) {
                alive_2 = group_allor<SPLIT>(alive_2);
                alive_3 = group_allor<SPLIT>(alive_3);
            }
        }
    }
#endif
    // Pair groups (DevModel::Group): the robot's pairs against one shape of another actor are skipped TOGETHER when that shape is
    // out of the robot's reach in every sample of this wavefront - one distance test per group and substep instead of a record
    // fetch, two pose reads and the broad-phase arithmetic per pair.  The verdict is wave-uniform (the pair loop is), so the masks
    // live in scalar registers; conservative like the broad phase (a skipped pair is one it would have culled).
    // (the kernel with a helper wavefront keeps TWO words: its 256-register budget is full, and the two scalar registers of the upper
    // words came out of it as 48 B more scratch per lane, pushing scene 1.19 -> 1.33 ms; mppi_create gives a scene with more than
    // kPairKernelMaxPairs candidate pairs the one-wavefront kernel instead)
    constexpr bool kWide = !kPair;
    unsigned dead_lo = 0u, dead_hi = 0u, dead_2 = 0u, dead_3 = 0u;
    if constexpr (kCached) {
        for (int g = 0; g < m.n_groups; g++) {
            const V3 d = shape_cached<T>(m, m.grp[g].anchor, L).p - shape_cached<T>(m, m.grp[g].other, L).p;
            bool far = dot(d, d) > m.grp[g].reach2;
#if defined(__HIP_DEVICE_COMPILE__)
            far = __all(far) != 0;
#endif
            if (far) {
                dead_lo |= m.grp[g].mask_lo; dead_hi |= m.grp[g].mask_hi;
                if constexpr (kWide) { dead_2 |= m.grp[g].mask_2; dead_3 |= m.grp[g].mask_3; }
            }
        }
#if defined(__HIP_DEVICE_COMPILE__)
        dead_lo = (unsigned)uniform((int)dead_lo);
        dead_hi = (unsigned)uniform((int)dead_hi);
        if constexpr (kWide) {
            dead_2 = (unsigned)uniform((int)dead_2);
            dead_3 = (unsigned)uniform((int)dead_3);
        }
#endif
        alive_lo &= ~dead_lo;
        alive_hi &= ~dead_hi;
        if constexpr (kWide) {
            alive_2 &= ~dead_2;
            alive_3 &= ~dead_3;
        }
    }
    auto is_dead = [&](int ip) MPPI_LAMBDA {
        if constexpr (kWide) {
            const unsigned wd = ip < 64 ? (ip < 32 ? dead_lo : dead_hi) : (ip < 96 ? dead_2 : dead_3);
            return ((wd >> (ip & 31)) & 1u) != 0u;
        } else {
            return (((ip < 32 ? dead_lo : dead_hi) >> (ip & 31)) & 1u) != 0u;
        }
    };
    MPPI_SEC(2);
    // The loop visits the pairs that are alive (all of them without the dealt pass).  Both 64-byte blocks of the NEXT pair
    // are requested while the current one is worked on: measured with the section clocks (tools/exp/section_clocks.py), a
    // quarter of the whole kernel was the exposed latency of this preamble - scalar loads of the pair record, then the
    // dependent LDS reads of the two shape poses, then the noise lookup behind two more dependent scalar loads.
    auto next_alive = [&](int from) MPPI_LAMBDA {
        if constexpr (kPair) {
#if defined(MPPI_PAIR_SPLIT)  // experiment builds (MPPI_BUILD_VARIANT=ps<k>): 1 = parities swapped, 2 = all pairs on the helper, 3 = none
            if (MPPI_PAIR_SPLIT == 1) return from + ((from ^ split.wave ^ 1) & 1);
            if (MPPI_PAIR_SPLIT == 2) return split.wave ? from : (int)m.n_pairs;
            if (MPPI_PAIR_SPLIT == 3) return split.wave ? (int)m.n_pairs : from;
#endif
            from += (from ^ split.wave) & 1;                        // pairs of this wavefront's parity ...
            while (from < m.n_pairs && is_dead(from)) from += 2;    // ... that no group verdict has removed
            return from < m.n_pairs ? from : (int)m.n_pairs;
            // (dealing the pairs that are left alternately in their ORDER instead - so that the split does not depend on which
            // groups are out of reach - measured 4 % slower on the pushing scene with every group in reach: 1.268 vs 1.221 ms)
        }
        if constexpr (dealt_broad_phase<T>(SPLIT)) {
            if (m.n_pairs > kDealtBroadPhaseMin) {
                const unsigned long long alive = (unsigned long long)alive_lo | ((unsigned long long)alive_hi << 32);
                const unsigned long long rest = from < 64 ? alive >> from : 0ull;
                if (rest != 0ull) return from + (int)__builtin_ctzll(rest);
                if (m.n_pairs > 64) {
                    const unsigned long long upper = (unsigned long long)alive_2 | ((unsigned long long)alive_3 << 32);
                    const int f2 = from > 64 ? from - 64 : 0;
                    const unsigned long long rest2 = f2 < 64 ? upper >> f2 : 0ull;
                    if (rest2 != 0ull) return 64 + f2 + (int)__builtin_ctzll(rest2);
                }
                return (int)m.n_pairs;
            }
        }
        while (from < m.n_pairs && is_dead(from)) from++;
        return from;
    };
    // (small trees walk all pairs in order: wave-uniform records, which the compiler would fetch with scalar loads - no
    // prefetch at all next to LDS work, see load_block_vmem; their integer fields go back to SGPRs for the scalar branches.
    // Measured alternatives at equal state, pushing scene: scalar loads requested one pair ahead 1.358 ms, vector loads one
    // pair ahead 1.324 ms, the records from the wavefront's LDS table one pair ahead 1.375 ms)
    // (round 4: the kernel with a helper wavefront as well.  Until its scratch traffic was removed in round 3 the 16 registers of
    // the NEXT pair's record did not fit its 256-register budget - 17 more values went to scratch; now they do: same scratch size,
    // 7 % fewer instructions in the listing, pushing scene 1.0625 -> 1.0499 ms at equal state, bit-identical costs)
    constexpr bool kVmemRecords = split_on_device(SPLIT) && !dealt_broad_phase<T>(SPLIT);
    // (the kernel with a helper wavefront runs on half the register file: the contact-law block of the pairs that survive the
    // broad phase is fetched when it is needed instead of occupying 16 registers across the whole pair)
    constexpr bool kLazyGains = kPair;
    auto fetch = [&](int i, PairGeom &g, PairGain &c) MPPI_LAMBDA {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (kVmemRecords) {
            g = load_block_vmem<PairGeom>(m.pr[i].g);
            if constexpr (!kLazyGains) c = load_block_vmem<PairGain>(m.pr[i].c);
            return;
        }
#endif
        g = load_block<PairGeom>(m.pr[i].g);
        c = load_block<PairGain>(m.pr[i].c);
    };
    int next = next_alive(0);
    PairGeom Gn;
    PairGain Cn_;
    if (next < m.n_pairs) fetch(next, Gn, Cn_);
    for (int ip = next; ip < m.n_pairs; ip = next) {
        PairGeom G = Gn;
        PairGain Cg = Cn_;
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (kVmemRecords) {
            G.a = uniform(G.a); G.b = uniform(G.b); G.mode = uniform(G.mode); G.rnd = uniform(G.rnd);
            G.typeA = uniform(G.typeA); G.typeB = uniform(G.typeB); G.entA = uniform(G.entA); G.entB = uniform(G.entB);
            G.rbA = uniform(G.rbA); G.rbB = uniform(G.rbB);
        }
#endif
        next = next_alive(ip + 1);
        if (next < m.n_pairs) fetch(next, Gn, Cn_);
        const bool has_b = G.b >= 0;
        ShapeW wa, wb;
        if constexpr (kCached) {
            wa = shape_cached<T>(m, G.a, L);
            if (has_b) wb = shape_cached<T>(m, G.b, L);
        } else {
            wa = shape_world(m.sh[G.a], root, L);
            if (has_b) wb = shape_world(m.sh[G.b], root, L);
        }
        float hA[3] = {G.hA[0], G.hA[1], G.hA[2]}, hB[3] = {G.hB[0], G.hB[1], G.hB[2]};
        ActorDraw da = {{0.f, 0.f, 0.f}, 1.f, 0.f}, db = {{0.f, 0.f, 0.f}, 1.f, 0.f};
        bool robotA = true, robotB = true;
        if (G.rnd) {  // this sample's own size of the noisy actors in the pair (friction and mass scale: see below)
            const int sa = ((G.rnd >> 8) & 15) - 1, sb = ((G.rnd >> 12) & 15) - 1;
            robotA = sa < 0;   // (here: "takes its nominal values")
            robotB = sb < 0;
            if (sa >= 0) {
                da = actor_draw_slot<T>(m, sa, 0.f, L);
                if (G.typeA == 0) for (int j = 0; j < 3; j++) hA[j] += 0.5f * da.d[j];
                else if (G.typeA == 1) hA[0] += da.d[0];
            }
            if (sb >= 0) {
                db = actor_draw_slot<T>(m, sb, 0.f, L);
                if (G.typeB == 0) for (int j = 0; j < 3; j++) hB[j] += 0.5f * db.d[j];
                else if (G.typeB == 1) hB[0] += db.d[0];
            }
        }
        const int typeA = G.typeA, typeB = G.typeB, rbB = G.rbB, entB = G.entB;
#if defined(MPPI_DUP) && defined(__HIP_DEVICE_COMPILE__)
        if constexpr (MPPI_DUP == 1 && kCached) {  // the preamble again: pair record from memory, both shape poses, noise draws
            const PairGeom G2 = load_block<PairGeom>(launder(&m)->pr[ip].g);
            const ShapeW w2 = shape_cached<T>(m, G2.a, L);
            float acc2 = w2.R.a[0] + w2.R.a[4] + w2.R.a[8] + w2.p.x + w2.p.y + w2.p.z + G2.hA[0];
            if (G2.b >= 0) {
                const ShapeW w3 = shape_cached<T>(m, G2.b, L);
                acc2 += w3.R.a[1] + w3.R.a[5] + w3.R.a[6] + w3.p.x + w3.p.y + w3.p.z;
            }
            if (G2.rnd) {
                const int sa2 = ((G2.rnd >> 8) & 15) - 1, sb2 = ((G2.rnd >> 12) & 15) - 1;
                if (sa2 >= 0) { const ActorDraw d2 = actor_draw_slot<T>(m, sa2, 0.f, L); acc2 += d2.d[0] + d2.d[1] + d2.d[2]; }
                if (sb2 >= 0) { const ActorDraw d2 = actor_draw_slot<T>(m, sb2, 0.f, L); acc2 += d2.d[0] + d2.d[1] + d2.d[2]; }
            }
            dup_keep(acc2);
        }
#endif
        MPPI_SEC(11);  // pair record, shape poses, per-sample sizes
        // Broad phase: six-axis SAT for two boxes, otherwise the bounding sphere of one shape against the other shape's
        // box grown by that radius (both ways).  Conservative by construction (a margin covers rounding), so skipping
        // changes no result; it removes the 2 x 26 feature-point tests of the many link-vs-table / link-vs-block
        // pairs that are nowhere near each other.
        bool apart = false;
        BoxRel rel;
        if (has_b && typeA == 0 && typeB == 0) {
            rel = box_relative(wa, wb);
            apart = boxes_apart(rel, hA, hB, 1e-4f);
        } else {
            constexpr float kMargin = 1e-4f;
            const float rA = typeA == 0 ? fsqrt(hA[0] * hA[0] + hA[1] * hA[1] + hA[2] * hA[2]) * 1.000001f : hA[0];  // (rounded up: conservative)
            if (!has_b) {
                // ground: a box by its support function along z (lowest corner at p.z - sum |R_zj| h_j: the chassis of a wheeled
                // base hovers within its bounding sphere's reach of the ground for ever), a sphere by its radius
                apart = typeA == 0 ? wa.p.z - (fabsf(wa.R.a[6]) * hA[0] + fabsf(wa.R.a[7]) * hA[1] + fabsf(wa.R.a[8]) * hA[2]) > kMargin
                                   : (typeA != 2 && wa.p.z > rA + kMargin);
            } else {   // (a disc against another shape: culled like a sphere of its radius)
                const float rB = typeB == 0 ? fsqrt(hB[0] * hB[0] + hB[1] * hB[1] + hB[2] * hB[2]) * 1.000001f : hB[0];
                const V3 d = wa.p - wb.p;
                if (typeB == 0) {
                    const V3 y = {wb.R.a[0] * d.x + wb.R.a[3] * d.y + wb.R.a[6] * d.z, wb.R.a[1] * d.x + wb.R.a[4] * d.y + wb.R.a[7] * d.z,
                                  wb.R.a[2] * d.x + wb.R.a[5] * d.y + wb.R.a[8] * d.z};
                    apart = fabsf(y.x) > hB[0] + rA + kMargin || fabsf(y.y) > hB[1] + rA + kMargin || fabsf(y.z) > hB[2] + rA + kMargin;
                } else {
                    apart = dot(d, d) > (rA + rB + kMargin) * (rA + rB + kMargin);
                }
                if (typeA == 0) {
                    const V3 x = {wa.R.a[0] * d.x + wa.R.a[3] * d.y + wa.R.a[6] * d.z, wa.R.a[1] * d.x + wa.R.a[4] * d.y + wa.R.a[7] * d.z,
                                  wa.R.a[2] * d.x + wa.R.a[5] * d.y + wa.R.a[8] * d.z};
                    apart = apart || fabsf(x.x) > hA[0] + rB + kMargin || fabsf(x.y) > hA[1] + rB + kMargin || fabsf(x.z) > hA[2] + rB + kMargin;
                }
            }
        }
#if !defined(__HIP_DEVICE_COMPILE__)
        {  // host builds (tests/hostemu): the restated test of the dealt broad phase must give the same verdict
            PairPose chk;
            if (pair_broad_phase<T, kCached>(m, ip, G, root, L, chk) != apart) {
                fprintf(stderr, "mppi_scene.hpp: pair_broad_phase disagrees with contact_forces on pair %d\n", ip);
                abort();
            }
        }
#endif
#if defined(MPPI_DUP) && defined(__HIP_DEVICE_COMPILE__)
        if constexpr (MPPI_DUP == 2) {
            PairPose pp2;
            dup_keep(pair_broad_phase<T, kCached>(m, ip, G, root, L, pp2) ? 1.f : 0.f);  // (poses + sizes + the SAT / sphere tests again)
        }
#endif
        MPPI_SEC(12);  // broad-phase arithmetic
        if (apart) continue;
        // contact law of the survivors: second block of the pair (already here: requested one pair ahead)
        // (requested through the vector memory path in front of the broad phase instead: measured the same, 1.0427 vs 1.0396 ms)
        if constexpr (kLazyGains) Cg = load_block<PairGain>(m.pr[ip].c);
        // (the kernel with a helper wavefront never sees a light body's pair - mppi_create - and carries none of that code: its
        // 256-register budget answered the run-time branches with 36 B more scratch per lane)
        Gains P = {kPair ? (G.mode < 3 ? G.mode : 0) : G.mode, Cg.mu, Cg.k, Cg.cn, Cg.ct, Cg.kh, Cg.inv_d0, Cg.npts};
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (kPair) __builtin_assume(P.mode >= 0 && P.mode <= 2);
#endif
        if (G.rnd) {  // per-sample friction (min of the two) and contact gains scaled with the per-sample reacting mass
            const float mua = robotA ? Cg.muA : da.mu;
            const float mub = !has_b ? Cg.mub : (robotB ? Cg.muB : db.mu);
            P.mu = fminf(mua, mub);
            const float ma = Cg.ma * da.ms, mb = Cg.mb * db.ms;
            const bool useA = G.mode == 1 || (!kPair && G.mode == 3);   // (3 / 4: the gains of the HEAVY body, the robot link)
            const float meff0 = G.mode == 0 ? Cg.ma * Cg.mb * frcp(Cg.ma + Cg.mb) : (useA ? Cg.ma : Cg.mb);
            const float meff = G.mode == 0 ? ma * mb * frcp(ma + mb) : (useA ? ma : mb);
            const float sc = meff * frcp(meff0);
            P.k *= sc; P.cn *= sc; P.ct *= sc; P.kh *= sc;
        }
        PairAcc acc;
        pair_zero(acc);
        if constexpr (kCached) {  // the cache holds poses only: velocities of the few pairs that get here
            wa.v = frame_velocity(L, G.entA);
            if (has_b) wb.v = frame_velocity(L, entB);
        }
        // a light body's pair: everything from here on in coordinates about that body's centre ("light bodies": its own inertia
        // must survive fp32 next to the heavy body's gains); the contact geometry is translation invariant
        const bool light = !kPair && G.mode >= 3;
        V3 lo = {0.f, 0.f, 0.f};
        if (light) {
            const int el = G.mode == 3 ? entB : G.entA;
            lo = V3{L[el * 18 + 9], L[el * 18 + 10], L[el * 18 + 11]};
            wa.p = wa.p - lo; wb.p = wb.p - lo;
            wa.v = light_motion_at(wa.v, lo); wb.v = light_motion_at(wb.v, lo);
        }
        // two dynamic boxes: one normal for the pair (box_pair_sat), unless the model asks for the law of ABI <= 7
        const bool dyn = G.mode == 0 || (!kPair && G.mode >= 3);
        const bool pair_normal = dyn && has_b && typeA == 0 && typeB == 0 && m.pair_normal != 0;
        BoxSat sat;
        sat.hit = false;
        if (pair_normal) sat = box_pair_sat(rel, wa, hA, wb, hB);
        // (two instantiations of the body: the pairs with ONE analytic point and the feature-point pairs - each call site below knows
        // which kind it has; compiled as one body at both sites the scene kernels carried every narrow phase twice: 21-23 % more
        // instructions in the listing, 68 instead of 48 B of scratch in the pushing scene's kernel, the same time within 0.6 %)
        constexpr bool kSplitBody = true;
        auto points = [&](auto single_tag, Split sp, PairAcc &out) MPPI_LAMBDA {
            constexpr bool kSingle = decltype(single_tag)::value, kMulti = !decltype(single_tag)::value;
            if (!has_b) {  // ground plane z = 0, normal +z (from ground to A)
                if (typeA == 0) {
                    if constexpr (kSplitBody && kSingle) return;
                    for (int c = sp.sub; c < 8; c += sp.n) {
                        V3 loc = {(c & 1) ? hA[0] : -hA[0], (c & 2) ? hA[1] : -hA[1], (c & 4) ? hA[2] : -hA[2]};
                        V3 pw = wa.p + mul(wa.R, loc);
                        if (pw.z < 0.f) contact_point_ground(P, pw, -pw.z, wa.v, out);
                    }
                } else if (sp.sub == 0) {
                    if constexpr (kSplitBody && kMulti) return;
                    if (typeA == 1) {
                        V3 pw = {wa.p.x, wa.p.y, wa.p.z - hA[0]};
                        if (pw.z < 0.f) contact_point_ground(P, pw, -pw.z, wa.v, out);
                    } else {  // disc: lowest point of the rim, axis = local z
                        V3 ax = {wa.R.a[2], wa.R.a[5], wa.R.a[8]};
                        V3 d = {ax.z * ax.x, ax.z * ax.y, ax.z * ax.z - 1.f};
                        float l2 = dot(d, d);
                        if (l2 > 1e-8f) {
                            V3 pw = wa.p + (hA[0] * frsqrt(l2)) * d;
                            if (pw.z < 0.f) contact_point_ground(P, pw, -pw.z, wa.v, out);
                        }
                    }
                }
            } else if (typeA == 0 && typeB == 0) {
                if constexpr (kSplitBody && kSingle) return;
                // A's points in B: centre t, columns of Rrel; B's points in A: centre -Rrel^T t, columns = rows of Rrel
                const V3 colA[3] = {{hA[0] * rel.R[0], hA[0] * rel.R[3], hA[0] * rel.R[6]}, {hA[1] * rel.R[1], hA[1] * rel.R[4], hA[1] * rel.R[7]},
                                    {hA[2] * rel.R[2], hA[2] * rel.R[5], hA[2] * rel.R[8]}};
                if (pair_normal) {
                    if (sat.hit) box_points_along(P, rel.t, colA, wb, hB, tmul3(wb.R, sat.n), sat.n, wa.v, wb.v, sp, out);
                } else {
                    box_points_in_box(P, rel.t, colA, wb, hB, 1.f, wa.v, wb.v, sp, out);
                }
                const V3 tb = {-(rel.R[0] * rel.t.x + rel.R[3] * rel.t.y + rel.R[6] * rel.t.z), -(rel.R[1] * rel.t.x + rel.R[4] * rel.t.y + rel.R[7] * rel.t.z),
                               -(rel.R[2] * rel.t.x + rel.R[5] * rel.t.y + rel.R[8] * rel.t.z)};
                const V3 colB[3] = {{hB[0] * rel.R[0], hB[0] * rel.R[1], hB[0] * rel.R[2]}, {hB[1] * rel.R[3], hB[1] * rel.R[4], hB[1] * rel.R[5]},
                                    {hB[2] * rel.R[6], hB[2] * rel.R[7], hB[2] * rel.R[8]}};
                if (pair_normal) {
                    if (sat.hit) box_points_along(P, tb, colB, wa, hA, -1.f * tmul3(wa.R, sat.n), sat.n, wa.v, wb.v, sp, out);
                } else {
                    box_points_in_box(P, tb, colB, wa, hA, -1.f, wa.v, wb.v, sp, out);
                }
            } else if (sp.sub == 0) {
                if constexpr (kSplitBody && kMulti) return;
                if (typeA == 1 && typeB == 0) sphere_in_box(P, wa.p, hA[0], wb, hB, 1.f, wa.v, wb.v, out);
                else if (typeA == 0 && typeB == 1) sphere_in_box(P, wb.p, hB[0], wa, hA, -1.f, wa.v, wb.v, out);
                else if (typeA == 1 && typeB == 1) sphere_sphere(P, wa.p, hA[0], wb.p, hB[0], wa.v, wb.v, out);
                else if (typeA == 2 && typeB == 0) disc_in_box(P, wa.p, V3{wa.R.a[2], wa.R.a[5], wa.R.a[8]}, hA[0], wb, hB, 1.f, wa.v, wb.v, out);
                else if (typeA == 0 && typeB == 2) disc_in_box(P, wb.p, V3{wb.R.a[2], wb.R.a[5], wb.R.a[8]}, hB[0], wa, hA, -1.f, wa.v, wb.v, out);
                else if (typeA == 2 && typeB == 1) disc_sphere(P, wa.p, V3{wa.R.a[2], wa.R.a[5], wa.R.a[8]}, hA[0], wb.p, hB[0], 1.f, wa.v, wb.v, out);
                else if (typeA == 1 && typeB == 2) disc_sphere(P, wb.p, V3{wb.R.a[2], wb.R.a[5], wb.R.a[8]}, hB[0], wa.p, hA[0], -1.f, wa.v, wb.v, out);
            }
        };
        MPPI_SEC(13);  // contact law, velocities
        // pairs with ONE analytic contact point (disc / sphere against the ground or a box): every lane of the sample computes
        // it - the lanes hold the same state, so the result is already identical in all of them and the cross-lane sum of the
        // pair's 31 partial values (more instructions than the contact itself) is not needed
        const bool single_point = has_b ? !(typeA == 0 && typeB == 0) : typeA != 0;
        if (single_point) {
            points(std::true_type{}, Split{0, 1}, acc);
        } else if constexpr (SPLIT == kSplitEmulate) {
            for (int sb = 0; sb < split.n; sb++) {
                PairAcc part;
                pair_zero(part);
                points(std::false_type{}, Split{sb, split.n}, part);
                pair_add(acc, part);
            }
        } else {
            points(std::false_type{}, split, acc);
#if defined(__HIP_DEVICE_COMPILE__)
            // (wave-uniform skip: most pairs are apart in most samples)
            if constexpr (split_on_device(SPLIT))
                if (__builtin_amdgcn_ballot_w64(acc.any) != 0) {
                    if (G.mode == 0) group_reduce_explicit<SPLIT>(acc);  // two dynamic bodies: no implicit damping block to sum
                    else group_reduce<SPLIT>(acc);
                    if (!kPair && G.mode >= 3) acc.wsum = group_allsum<SPLIT>(acc.wsum);   // (a light body's pair: patch weights as well)
                }
#endif
        }
#if defined(MPPI_DUP) && defined(__HIP_DEVICE_COMPILE__)
        if constexpr (MPPI_DUP == 3) {  // the feature points / analytic contact point of the pair again (no cross-lane sum)
            PairAcc a2;
            pair_zero(a2);
            if (single_point) points(std::true_type{}, Split{0, 1}, a2); else points(std::false_type{}, split, a2);
            dup_keep(a2.f.a.x + a2.f.a.y + a2.f.a.z + a2.f.l.x + a2.f.l.y + a2.f.l.z + a2.rep.x + a2.C.I.xx + a2.C.I.yz + a2.C.M.zz + a2.C.H[1] + a2.C.H[5] + a2.C.H[6]);
        }
        if constexpr (MPPI_DUP == 4 && split_on_device(SPLIT)) {  // the cross-lane sum of a contact-bearing pair again
            if (!single_point && __builtin_amdgcn_ballot_w64(acc.any) != 0) {
                PairAcc a2 = acc;
                if (G.mode == 0) group_reduce_explicit<SPLIT>(a2);
                else group_reduce<SPLIT>(a2);
                dup_keep(a2.f.a.x + a2.f.l.z + a2.rep.y + a2.C.I.xy + a2.C.M.xx + a2.C.H[3]);
            }
        }
#endif
        MPPI_SEC(14);  // feature points + cross-lane sum
        if (pair_normal && sat.hit) box_pair_fill(P, rel, wa, hA, wb, hB, sat.n, acc);
        if (dyn && acc.any) pair_normalise(P, acc);
        if (light && acc.any) {
            // ("light bodies"; everything here is about the light body's centre `lo`)
            //   robot link X:  -+f + C v_L(start) and C, shifted to the world origin, into the link's rows like a static contact's;
            //   light body L:  +-f + C v_X(start) and C into the row of the light region; the link's parent frame competes for the
            //                  reference frame (nearest the base wins), (C S) of the link's own joint goes into the link's record
            const bool heavyA = G.mode == 3;
            const int eh = heavyA ? G.entA : entB;
            const SV fA = acc.f, fB = {{-acc.f.a.x, -acc.f.a.y, -acc.f.a.z}, {-acc.f.l.x, -acc.f.l.y, -acc.f.l.z}};
            const SV CvL = mul(acc.C, heavyA ? wb.v : wa.v), CvX = mul(acc.C, heavyA ? wa.v : wb.v);
            const SV fh = light_shift((heavyA ? fA : fB) + CvL, lo), fl = (heavyA ? fB : fA) + CvX;
            const AI Ch = light_shift(acc.C, lo);
            const float rec[kLightRow] = {fl.a.x, fl.a.y, fl.a.z, fl.l.x, fl.l.y, fl.l.z, acc.C.I.xx, acc.C.I.xy, acc.C.I.xz, acc.C.I.yy, acc.C.I.yz, acc.C.I.zz,
                                          acc.C.H[0], acc.C.H[1], acc.C.H[2], acc.C.H[3], acc.C.H[4], acc.C.H[5], acc.C.H[6], acc.C.H[7], acc.C.H[8],
                                          acc.C.M.xx, acc.C.M.xy, acc.C.M.xz, acc.C.M.yy, acc.C.M.yz, acc.C.M.zz};
            const int ocf = kCfW + 3 * G.rbA, ob = kCfW + 3 * rbB;
            cf_touched |= (1u << (G.rbA & 31)) | (1u << (rbB & 31));
            touched |= 1u << eh;
            // the link's own joint about lo, and (the pair's C') S; its parent frame (a shape on a moving base: the base itself)
            constexpr int NBl = T::NB;
            const bool on_body = eh < NBl;
            const int ehb = on_body ? eh : 0;
            const V3 az = {L[ehb * 18 + 2], L[ehb * 18 + 5], L[ehb * 18 + 8]}, pl = V3{L[ehb * 18 + 9], L[ehb * 18 + 10], L[ehb * 18 + 11]} - lo;
            const SV Sj = m.b[ehb].k0.jtype == 0 ? SV{az, cross(pl, az)} : SV{{0.f, 0.f, 0.f}, az};
            const SV gj = mul(acc.C, Sj);
            const int parj = T::par[ehb];
            const int par = !on_body ? eh : (parj < 0 ? NBl + (-1 - parj) : parj);
            auto key = [](int f) MPPI_LAMBDA { return f < NBl ? f + T::NBASE : f - NBl; };   // (bases first, then the bodies in their order)
            const bool leader = !split_on_device(SPLIT) || split.sub == 0;
            // (kernels whose lanes share a sample: the leader adds inside the LDS unit - ds_add_f32, no read - wait - add - write round trip
            // per word; 27 + 27 + 6 + 6 words per active light pair, four pairs on a held block: the read-modify-write form was 13 % of
            // the gripper scene's kernel at the `held` state)
            constexpr bool kLdsAdd = split_on_device(SPLIT);
            auto add_to = [&](float &x, float v) MPPI_LAMBDA {
#if defined(__HIP_DEVICE_COMPILE__)
                if constexpr (kLdsAdd) { lds_add(x, v); return; }
#endif
                x += v;
            };
            if (leader) {
                for (int j = 0; j < kLightRow; j++) add_to(L.lt(j), rec[j]);
                const int ref = __builtin_bit_cast(int, L.lt(kLightRef));
                if (ref < 0 || key(par) < key(ref)) L.lt(kLightRef) = __builtin_bit_cast(float, par);
                if (on_body) {
                    int k = 0;
                    while (k < kLightSlots) {
                        const int jk = __builtin_bit_cast(int, L.lt(kLightRec + k * kLightSlotFloats));
                        if (jk == eh || jk < 0) break;
                        k++;
                    }
                    if (k < kLightSlots) {   // (a fifth link in contact in one substep: its own joint's change is not passed on)
                        const int o = kLightRec + k * kLightSlotFloats;
                        const bool fresh = __builtin_bit_cast(int, L.lt(o)) < 0;
                        L.lt(o) = __builtin_bit_cast(float, eh);
                        const float g6[6] = {gj.a.x, gj.a.y, gj.a.z, gj.l.x, gj.l.y, gj.l.z};
                        if (fresh) for (int j = 0; j < 6; j++) L.lt(o + 1 + j) = g6[j];
                        else for (int j = 0; j < 6; j++) add_to(L.lt(o + 1 + j), g6[j]);
                    }
                }
                const int oa = kAccW + eh * 27;
                const float row[27] = {fh.a.x, fh.a.y, fh.a.z, fh.l.x, fh.l.y, fh.l.z, Ch.I.xx, Ch.I.xy, Ch.I.xz, Ch.I.yy, Ch.I.yz, Ch.I.zz,
                                       Ch.H[0], Ch.H[1], Ch.H[2], Ch.H[3], Ch.H[4], Ch.H[5], Ch.H[6], Ch.H[7], Ch.H[8],
                                       Ch.M.xx, Ch.M.xy, Ch.M.xz, Ch.M.yy, Ch.M.yz, Ch.M.zz};
                for (int j = 0; j < 27; j++) add_to(L[oa + j], row[j]);   // (= acc_add(L, kAccW, eh, fh, &Ch))
                add_to(L[ocf], acc.rep.x); add_to(L[ocf + 1], acc.rep.y); add_to(L[ocf + 2], acc.rep.z);
                add_to(L[ob], -acc.rep.x); add_to(L[ob + 1], -acc.rep.y); add_to(L[ob + 2], -acc.rep.z);
            }
            n_light++;
        } else
        if (acc.any) {
            const SV neg = {{-acc.f.a.x, -acc.f.a.y, -acc.f.a.z}, {-acc.f.l.x, -acc.f.l.y, -acc.f.l.z}};
            const int ocf = kCfW + 3 * G.rbA, ob = kCfW + 3 * (rbB >= 0 ? rbB : 0);
            cf_touched |= (1u << (G.rbA & 31)) | (rbB >= 0 ? 1u << (rbB & 31) : 0u);
            touched |= G.mode == 0 ? (1u << G.entA) | (1u << entB) : (G.mode == 1 ? 1u << G.entA : 1u << entB);
#if defined(__HIP_DEVICE_COMPILE__)
            if constexpr (split_on_device(SPLIT)) {
                const bool leader = split.sub == 0;
                if (G.mode == 0) {
                    acc_add_shared(L, kAccW, G.entA, acc.f, nullptr, leader);
                    acc_add_shared(L, kAccW, entB, neg, nullptr, leader);
                } else {
                    acc_add_shared(L, kAccW, G.mode == 1 ? G.entA : entB, G.mode == 1 ? acc.f : neg, &acc.C, leader);
                }
                if (leader) {
                    lds_add(L[ocf], acc.rep.x); lds_add(L[ocf + 1], acc.rep.y); lds_add(L[ocf + 2], acc.rep.z);
                    if (rbB >= 0) { lds_add(L[ob], -acc.rep.x); lds_add(L[ob + 1], -acc.rep.y); lds_add(L[ob + 2], -acc.rep.z); }
                }
            } else
#endif
            {
                if (G.mode == 0) {
                    acc_add(L, Lay::kAcc, G.entA, acc.f, nullptr);
                    acc_add(L, Lay::kAcc, entB, neg, nullptr);
                } else if (G.mode == 1) {
                    acc_add(L, Lay::kAcc, G.entA, acc.f, &acc.C);
                } else {
                    acc_add(L, Lay::kAcc, entB, neg, &acc.C);
                }
                L[ocf] += acc.rep.x; L[ocf + 1] += acc.rep.y; L[ocf + 2] += acc.rep.z;
                if (rbB >= 0) { L[ob] -= acc.rep.x; L[ob + 1] -= acc.rep.y; L[ob + 2] -= acc.rep.z; }
            }
        }
#if defined(MPPI_DUP) && defined(__HIP_DEVICE_COMPILE__)
        if constexpr (MPPI_DUP == 5) {  // the same read-modify-write traffic on the accumulator rows with zeros
            if (acc.any) {
                PairAcc z2;
                pair_zero(z2);
                z2.f.a.x = launder_zero();
                if (G.mode == 0) { acc_add(L, Lay::kAcc, G.entA, z2.f, nullptr); acc_add(L, Lay::kAcc, entB, z2.f, nullptr); }
                else acc_add(L, Lay::kAcc, G.mode == 1 ? G.entA : entB, z2.f, &z2.C);
            }
        }
#endif
        MPPI_SEC(15);  // accumulate into the frames' LDS rows
    }
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (kPair) {
        if (helper && split.sub == 0) {
            L[L.xch] = __builtin_bit_cast(float, touched);
            L[L.xch + 1] = __builtin_bit_cast(float, cf_touched);
        }
        MPPI_BARRIER(3);  // both halves of the pair list are accumulated
        if (!helper) {
            // set 0 += set 1, rows the helper wrote: each of the sample's lanes takes every 8th value (one lane per address,
            // LDS operations of a wavefront execute in order: the readers below see the sums)
            const unsigned t1 = __builtin_bit_cast(unsigned, L[L.xch]), c1 = __builtin_bit_cast(unsigned, L[L.xch + 1]);
            // (the rows of the free actors stay apart: the helper solves those bodies itself from both sets, free_body_accel)
            for (int e = 0; e <= T::NB; e++)
                if ((t1 >> e) & 1u)
                    for (int j = split.sub; j < 27; j += split.n) L[Lay::kAcc + 27 * e + j] += L[L.set1 + 27 * e + j];
            const int cf1 = L.set1 + (Lay::kCf - Lay::kAcc);
            for (int j = split.sub; j < 3 * m.n_rb; j += split.n)
                if (cf_all || ((c1 >> (j / 3)) & 1u)) L[Lay::kCf + j] += L[cf1 + j];
            touched |= t1 & ((2u << T::NB) - 1u);
            cf_touched |= c1;
        }
    }
#endif
    acc_dirty = touched;
    cf_dirty = cf_touched;
    MPPI_SEC(3);
    return touched;
}

// quaternion (xyzw) integration q <- normalize(q + h/2 (w,0) * q), world-frame angular velocity
MPPI_HD void quat_integrate(float *q, V3 w, float h) {
    const float x = q[0], y = q[1], z = q[2], s = q[3];
    const float k = 0.5f * h;
    float nx = x + k * (w.x * s + w.y * z - w.z * y);
    float ny = y + k * (w.y * s + w.z * x - w.x * z);
    float nz = z + k * (w.z * s + w.x * y - w.y * x);
    float ns = s - k * (w.x * x + w.y * y + w.z * z);
    const float inv = frsqrt(nx * nx + ny * ny + nz * nz + ns * ns);
    q[0] = nx * inv; q[1] = ny * inv; q[2] = nz * inv; q[3] = ns * inv;
}

// semi-implicit Euler of a 13-float root row under the world-origin spatial acceleration a
MPPI_HD void root_integrate(float *rs, const SV &a, float h) {
    V3 p = loadv(rs), v = loadv(rs + 7), w = loadv(rs + 10);
    // acceleration of the body origin: aO + alpha x p + w x v
    V3 vd = a.l + cross(a.a, p) + cross(w, v);
    w = w + h * a.a;
    v = v + h * vd;
    p = p + h * v;
    quat_integrate(rs + 3, w, h);
    rs[0] = p.x; rs[1] = p.y; rs[2] = p.z;
    rs[7] = v.x; rs[8] = v.y; rs[9] = v.z;
    rs[10] = w.x; rs[11] = w.y; rs[12] = w.z;
}

// ... of a FREE actor: and its angular velocity limited to MPPI_MAX_ANGULAR_VELOCITY (Isaac Gym's AssetOptions default, include/mppi_hip.h:
// a one-gram block that a gripper has squeezed out like a seed would otherwise turn by more than a radian per substep)
MPPI_HD void free_integrate(float *rs, const SV &a, float h) {
    root_integrate(rs, a, h);
    const V3 w = loadv(rs + 10);
    const float w2 = dot(w, w);
    constexpr float wm = (float)MPPI_MAX_ANGULAR_VELOCITY;
    if (w2 > wm * wm) {
        const float sc = wm * frsqrt(w2);
        rs[10] = sc * w.x; rs[11] = sc * w.y; rs[12] = sc * w.z;
    }
}

// spatial velocity / acceleration of the bases of the forest (one: every model but an env of several moving-base robots)
template <class T>
struct BaseSV {
    SV r[T::NBASE];
};
// Articulated-body solve of the robot with external wrenches f_i and implicit dampings C_i per frame
// (from contact_forces), explicit gravity, optional floating base - one 6x6 base system per tree of a forest of moving-base
// robots (a root body's parent -1 - r names its base, mppi_device.hpp Topo).  Returns qdd and the base accelerations.
template <class T, class M>
MPPI_HD void aba_scene(M &m, const Pose<T> &P, const BaseSV<T> &vbase, const float *qd, const float *tau_exp, const float *kdh,
                       const LMem &L, float *qdd, BaseSV<T> &abase) {
    constexpr int NB = T::NB;
    using Lay = SceneLayout<T>;
    constexpr int NBs = NB ? NB : 1;
    constexpr int NX = T::NBASE;  // bases: accumulators NB .. NB + NX - 1
    SV v[NBs], U[NBs], pacc[NBs + NX];
    AI acc[NBs + NX];
    float invd[NBs], u[NBs];
    bool has_acc[NBs + NX];
    const float h = m.h;
    const V3 g = m.gravity_on ? V3{m.g[0], m.g[1], m.g[2]} : V3{0.f, 0.f, 0.f};
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        SV S = joint_subspace<T, i>(m, P);
        SV sj = {qd[i] * S.a, qd[i] * S.l};
        if constexpr (par < 0) v[i] = vbase.r[base_of_parent(par)] + sj;
        else v[i] = v[par < 0 ? 0 : par] + sj;
        has_acc[i] = false;
    });
    static_for<0, NX>([&](auto rc) MPPI_LAMBDA { has_acc[NB + rc] = false; });
    static_rfor<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const BodyK1 b = load_block<BodyK1>(m.b[i].k1);
        SV S = joint_subspace<T, i>(m, P);
        AI A;
        SV pA;
        V3 hw;
        const float Ic6[6] = {b.Ic(0), b.Ic(1), b.Ic(2), b.Ic(3), b.Ic(4), b.Ic(5)};
        rigid_world(P.R[i], P.p[i], b.m, V3{b.hb(0), b.hb(1), b.hb(2)}, Ic6, v[i], A, pA, hw);
        // gravity, contact wrench, implicit contact damping:  (IA + h C) a + (pA + C v - f - f_g) = 0
        SV fe;
        AI C;
        acc_load(L, Lay::kAcc, i, fe, C);
        SV Cv = mul(C, v[i]);
        pA = {pA.a + Cv.a - fe.a - cross(hw, g), pA.l + Cv.l - fe.l - b.m * g};
        A.I.xx += h * C.I.xx; A.I.xy += h * C.I.xy; A.I.xz += h * C.I.xz; A.I.yy += h * C.I.yy; A.I.yz += h * C.I.yz; A.I.zz += h * C.I.zz;
        for (int j = 0; j < 9; j++) A.H[j] += h * C.H[j];
        A.M.xx += h * C.M.xx; A.M.xy += h * C.M.xy; A.M.xz += h * C.M.xz; A.M.yy += h * C.M.yy; A.M.yz += h * C.M.yz; A.M.zz += h * C.M.zz;
        if (has_acc[i]) {
            add_to(A, acc[i]);
            pA = pA + pacc[i];
        }
        U[i] = mul(A, S);
        float d = dot(S, U[i]) + kdh[i];
        invd[i] = frcp(d);
        u[i] = tau_exp[i] - dot(S, pA);
        const SV vp = par < 0 ? vbase.r[base_of_parent(par)] : v[par < 0 ? 0 : par];
        SV sj = {qd[i] * S.a, qd[i] * S.l};
        SV c = {cross(vp.a, sj.a), cross(vp.a, sj.l) + cross(vp.l, sj.a)};
        SV Ic_ = mul(A, c);
        float k = (u[i] - dot(U[i], c)) * invd[i];
        SV pa = {pA.a + Ic_.a + k * U[i].a, pA.l + Ic_.l + k * U[i].l};
        rank1_sub(A, U[i], invd[i]);
        constexpr int pj = par < 0 ? NB + base_of_parent(par) : par;  // base accumulators at NB ..
        if (has_acc[pj]) {
            add_to(acc[pj], A);
            pacc[pj] = pacc[pj] + pa;
        } else {
            acc[pj] = A;
            pacc[pj] = pa;
            has_acc[pj] = true;
        }
    });
    static_for<0, NX>([&](auto rc) MPPI_LAMBDA { abase.r[rc] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}; });
    if (m.floating) {
        static_for<0, NX>([&](auto rc) MPPI_LAMBDA {
            constexpr int r = rc;
            AI A;
            SV pA;
            V3 hw;
            const float bm = base_mass<r>(m);
            if constexpr (r == 0) rigid_world(P.Rb, P.pb, bm, loadv(m.base_hb), m.base_Ic, vbase.r[0], A, pA, hw);
            else rigid_world(P.Rx[r - 1], P.px[r - 1], bm, loadv(m.xbase_hb[r - 1]), m.xbase_Ic[r - 1], vbase.r[r], A, pA, hw);
            SV fe;
            AI C;
            acc_load(L, Lay::kAcc, NB + r, fe, C);
            SV Cv = mul(C, vbase.r[r]);
            pA = {pA.a + Cv.a - fe.a - cross(hw, g), pA.l + Cv.l - fe.l - bm * g};
            A.I.xx += h * C.I.xx; A.I.xy += h * C.I.xy; A.I.xz += h * C.I.xz; A.I.yy += h * C.I.yy; A.I.yz += h * C.I.yz; A.I.zz += h * C.I.zz;
            for (int j = 0; j < 9; j++) A.H[j] += h * C.H[j];
            A.M.xx += h * C.M.xx; A.M.xy += h * C.M.xy; A.M.xz += h * C.M.xz; A.M.yy += h * C.M.yy; A.M.yz += h * C.M.yz; A.M.zz += h * C.M.zz;
            if (has_acc[NB + r]) {
                add_to(A, acc[NB + r]);
                pA = pA + pacc[NB + r];
            }
            SV rhs = {{-pA.a.x, -pA.a.y, -pA.a.z}, {-pA.l.x, -pA.l.y, -pA.l.z}};
            abase.r[r] = solve6(A, rhs);
        });
    }
    SV a[NBs];
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        SV S = joint_subspace<T, i>(m, P);
        const SV vp = par < 0 ? vbase.r[base_of_parent(par)] : v[par < 0 ? 0 : par];
        SV sj = {qd[i] * S.a, qd[i] * S.l};
        SV c = {cross(vp.a, sj.a), cross(vp.a, sj.l) + cross(vp.l, sj.a)};
        SV ap = (par < 0 ? abase.r[base_of_parent(par)] : a[par < 0 ? 0 : par]) + c;
        float dd = (u[i] - dot(U[i], ap)) * invd[i];
        qdd[i] = dd;
        a[i] = {ap.a + dd * S.a, ap.l + dd * S.l};
    });
}

// kinematics of the whole scene for the current state: robot poses + dynamic frames into L
template <class T, class M>
MPPI_HD void scene_frames(M &m, const float *root, const SceneState<T> &s, Pose<T> &P, BaseSV<T> &vbase, const LMem &L) {
    constexpr int NB = T::NB;
    // the robot rows of `root` are replaced by the sample's own base states
    static_for<0, T::NBASE>([&](auto rc) MPPI_LAMBDA {
        constexpr int r = rc;
        const float *bs = s.template base_row<r>();
        P.template base_p<r>() = loadv(bs);
        P.template base_R<r>() = quat_to_R(bs + 3);
    });
    forward_kinematics_base<T>(m, s.q, P);
    static_for<0, T::NBASE>([&](auto rc) MPPI_LAMBDA {
        constexpr int r = rc;
        const float *bs = s.template base_row<r>();
        V3 wb = loadv(bs + 10), vb = loadv(bs + 7);
        vbase.r[r] = m.floating ? SV{wb, vb - cross(wb, P.template base_p<r>())} : SV{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    });
    SV v[NB ? NB : 1];
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        SV S = joint_subspace<T, i>(m, P);
        SV sj = {s.qd[i] * S.a, s.qd[i] * S.l};
        if constexpr (par < 0) v[i] = vbase.r[base_of_parent(par)] + sj;
        else v[i] = v[par < 0 ? 0 : par] + sj;
        frame_store(L, i, P.R[i], P.p[i], v[i]);
    });
    static_for<0, T::NBASE>([&](auto rc) MPPI_LAMBDA {
        constexpr int r = rc;
        frame_store(L, NB + r, P.template base_R<r>(), P.template base_p<r>(), vbase.r[r]);
    });
    for (int f = 0; f < kFreeSlots; f++)
        if (f < m.n_free) {
            const float *rs = s.fr[f];
            V3 p = loadv(rs), w = loadv(rs + 10), vl = loadv(rs + 7);
            frame_store(L, free_frame<T>(f), quat_to_R(rs + 3), p, SV{w, vl - cross(w, p)});
        }
}

// free rigid bodies (frames and accumulators of this substep in L): (I + h C) a = -(v x* I v + C v - f - f_g)
// `set1` > 0: the accumulator rows are the sum of the owner's set and the helper wavefront's set at that row offset
// (kSplitOctPair: the helper solves the free actors while the owner solves the robot)
// LIGHT: the kernel can have light bodies (never the one with a helper wavefront, whose helper calls this with LIGHT = false: its
// 256-register budget carries none of that code)
template <class T, bool LIGHT = true, class M>
MPPI_HD SV free_body_accel(M &m, int f, const LMem &L, float h, int set1 = 0, const SV *dv_ref = nullptr, const float *dqd = nullptr) {
    constexpr int NB = T::NB;
    auto &F = m.fr[f];
    M3 R;
    V3 p;
    SV v;
    frame_load(L, free_frame<T>(f), R, p, v);
    float fm = F.m, Ic6[6] = {F.Ic[0], 0.f, 0.f, F.Ic[1], 0.f, F.Ic[2]};
    if (m.rnd_slot[F.actor] >= 0) {  // this sample's own mass and size
        const ActorDraw dr = actor_draw<T>(m, F.actor, L);
        fm *= dr.ms;
        if (F.type == 1) {  // MPPI_ACTOR_BOX
            const float x = F.size[0] + dr.d[0], y = F.size[1] + dr.d[1], z = F.size[2] + dr.d[2];
            const float m12 = fm * (1.f / 12.f);
            Ic6[0] = m12 * (y * y + z * z); Ic6[3] = m12 * (x * x + z * z); Ic6[5] = m12 * (x * x + y * y);
        } else {  // sphere
            const float r = F.size[0] + dr.d[0];
            Ic6[0] = Ic6[3] = Ic6[5] = 0.4f * fm * r * r;
        }
    }
    AI A;
    SV pA;
    V3 hw;
    // (a light body that no robot link touches in this substep - no reference frame recorded, contact_forces - is an ordinary free
    // body: the branch below exists for the robot's gains next to a gram's inertia)
    if (LIGHT && dv_ref != nullptr && m.n_light_pairs != 0 && ((m.light_free >> f) & 1u) != 0u && L.lp != nullptr && __builtin_bit_cast(int, L.lt(kLightRef)) >= 0) {
        // the LIGHT body ("light bodies" above), solved about its own centre o = p:
        //   (I + h (C_s + C')) a = -(v x* I v + (C_s + C') v - f_s - f' - f_g) + C' dv_ref + sum_X (C' S)_X dqd_X
        // C_s, f_s: its contacts with static geometry (the accumulator rows, shifted from the world origin to o); C', f': its pairs with
        // robot links (the light region's row: already about o, f' holds +-f + C v_X(start)); the links' velocity changes over the substep
        // (dv_ref, dqd: step_free_bodies) enter through the reference frame and the links' records
        const SV vo = light_motion_at(v, p);
        rigid_world(R, V3{0.f, 0.f, 0.f}, fm, V3{0.f, 0.f, 0.f}, Ic6, vo, A, pA, hw);
        SV fs;
        AI Cs;
        acc_load(L, SceneLayout<T>::kAcc, free_frame<T>(f), fs, Cs);
        const V3 mo = {-p.x, -p.y, -p.z};
        SV fe = light_shift(fs, mo);
        AI C = light_shift(Cs, mo);
        AI Cr;
        Cr.I = {L.lt(6), L.lt(7), L.lt(8), L.lt(9), L.lt(10), L.lt(11)};
        for (int j = 0; j < 9; j++) Cr.H[j] = L.lt(12 + j);
        Cr.M = {L.lt(21), L.lt(22), L.lt(23), L.lt(24), L.lt(25), L.lt(26)};
        add_to(C, Cr);
        fe = {fe.a + V3{L.lt(0), L.lt(1), L.lt(2)}, fe.l + V3{L.lt(3), L.lt(4), L.lt(5)}};
        // the links' velocity changes over the substep: (sum C') dv_ref + sum_links (C' S)_link dqd_link  (oracle light_pair_t)
        {   // (dv_ref: light_reference_change, about the world origin)
            const SV d = mul(Cr, light_motion_at(*dv_ref, p));
            fe = {fe.a + d.a, fe.l + d.l};
        }
        for (int sl = 0; sl < kLightSlots; sl++) {
            const int o = kLightRec + sl * kLightSlotFloats;
            const int j = __builtin_bit_cast(int, L.lt(o));
            if (j < 0) break;
            float dq = 0.f;   // the rate change of the link's own joint (a select over the tree's joints: j comes out of the record)
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA { dq = j == (int)ic ? dqd[ic] : dq; });
            fe = {fe.a + dq * V3{L.lt(o + 1), L.lt(o + 2), L.lt(o + 3)}, fe.l + dq * V3{L.lt(o + 4), L.lt(o + 5), L.lt(o + 6)}};
        }
        const V3 g = F.gravity ? V3{m.g[0], m.g[1], m.g[2]} : V3{0.f, 0.f, 0.f};
        const SV Cv = mul(C, vo);
        pA = {pA.a + Cv.a - fe.a, pA.l + Cv.l - fe.l - fm * g};
        A.I.xx += h * C.I.xx; A.I.xy += h * C.I.xy; A.I.xz += h * C.I.xz; A.I.yy += h * C.I.yy; A.I.yz += h * C.I.yz; A.I.zz += h * C.I.zz;
        for (int j = 0; j < 9; j++) A.H[j] += h * C.H[j];
        A.M.xx += h * C.M.xx; A.M.xy += h * C.M.xy; A.M.xz += h * C.M.xz; A.M.yy += h * C.M.yy; A.M.yz += h * C.M.yz; A.M.zz += h * C.M.zz;
        const SV rhs = {{-pA.a.x, -pA.a.y, -pA.a.z}, {-pA.l.x, -pA.l.y, -pA.l.z}};
        const SV ao = solve6(A, rhs);
        return SV{ao.a, ao.l - cross(ao.a, p)};   // (about the world origin, what root_integrate takes)
    }
    rigid_world(R, p, fm, V3{0.f, 0.f, 0.f}, Ic6, v, A, pA, hw);
    SV fe;
    AI C;
    acc_load(L, SceneLayout<T>::kAcc, free_frame<T>(f), fe, C);
    if (set1 > 0) {  // (fixed order: owner's set + helper's set, as the owner's merge does for the robot's rows)
        SV fe1;
        AI C1;
        acc_load(L, set1, free_frame<T>(f), fe1, C1);
        fe = fe + fe1;
        add_to(C, C1);
    }
    const V3 g = F.gravity ? V3{m.g[0], m.g[1], m.g[2]} : V3{0.f, 0.f, 0.f};
    SV Cv = mul(C, v);
    pA = {pA.a + Cv.a - fe.a - cross(hw, g), pA.l + Cv.l - fe.l - fm * g};
    A.I.xx += h * C.I.xx; A.I.xy += h * C.I.xy; A.I.xz += h * C.I.xz; A.I.yy += h * C.I.yy; A.I.yz += h * C.I.yz; A.I.zz += h * C.I.zz;
    for (int j = 0; j < 9; j++) A.H[j] += h * C.H[j];
    A.M.xx += h * C.M.xx; A.M.xy += h * C.M.xy; A.M.xz += h * C.M.xz; A.M.yy += h * C.M.yy; A.M.yz += h * C.M.yz; A.M.zz += h * C.M.zz;
    SV rhs = {{-pA.a.x, -pA.a.y, -pA.a.z}, {-pA.l.x, -pA.l.y, -pA.l.z}};
    return solve6(A, rhs);
}
#if defined(__HIP_DEVICE_COMPILE__)
// helper wavefront of kSplitOctPair, after its half of the pairs: the free actors' accelerations from both accumulator sets,
// handed to the owner through the sample's LDS row; the barrier pairs with the owner's in step_scene
template <class T, class M>
__device__ __forceinline__ void helper_free_bodies(M &m, const LMem &L, Split split) {
    const float h = m.h;
    for (int f = 0; f < kFreeSlots; f++)
        if (f < m.n_free) {
            SV a = free_body_accel<T, false>(m, f, L, h, L.set1);
            {   // the angular-velocity limit of free actors (free_integrate) HERE, as the acceleration that lands on it: the owner, full at 256
                // registers, integrates with the plain root_integrate (the limit inside its integration cost it 24 B of scratch per lane).
                // alpha' = (w_limited - w) / h;  the body origin's acceleration a_O + alpha x p + w x v stays what it was
                const int o = free_frame<T>(f) * 18;
                const V3 p = {L[o + 9], L[o + 10], L[o + 11]}, w = {L[o + 12], L[o + 13], L[o + 14]};
                const V3 w1 = w + h * a.a;
                const float w2 = dot(w1, w1);
                constexpr float wm = (float)MPPI_MAX_ANGULAR_VELOCITY;
                if (w2 > wm * wm) {
                    const V3 al = (wm * frsqrt(w2)) * w1 - w;
                    const V3 an = frcp(h) * al;
                    a = SV{an, a.l + cross(a.a - an, p)};
                }
            }
            if (split.sub == 0) {
                const int o = L.xch + 2 + 6 * f;
                L[o] = a.a.x; L[o + 1] = a.a.y; L[o + 2] = a.a.z; L[o + 3] = a.l.x; L[o + 4] = a.l.y; L[o + 5] = a.l.z;
            }
        }
    MPPI_BARRIER(4);
}
#endif
// dqd: how far the joint rates changed over the substep (new - old, after the limits; the three step functions hand it over) - a light
// body's solve wants the velocity changes of the links that touch it ("light bodies")
template <class T, class M>
MPPI_HD void step_free_bodies(M &m, SceneState<T> &s, const LMem &L, float h, const float *dqd) {
    // (only a sample whose light body met a link in this substep - a reference frame is recorded then, contact_forces - pays for any of
    // it: the gripper scene at 65 536 samples, most of them nowhere near the block, paid 128 -> 101 Hz before)
    SV dv_ref = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    const int ref = m.n_light_pairs != 0 && L.lp != nullptr ? __builtin_bit_cast(int, L.lt(kLightRef)) : -1;
    if (ref >= 0) dv_ref = light_reference_change<T>(m, s, L, dqd, ref);
    for (int f = 0; f < kFreeSlots; f++)
        if (f < m.n_free) free_integrate(s.fr[f], free_body_accel<T>(m, f, L, h, 0, &dv_ref, dqd), h);
}

// The kernel with a helper wavefront lives on half the register file (256 registers), and the candidate-pair loop alone wants
// ~170 of them: whatever else is alive across it goes to scratch memory - the compiler parked 19 registers around the loop of
// every substep and 79 in all (~250 MB of scratch write-back per launch at K = 8192, twelve times the algorithmic traffic).
// The sample's state (q, qd, base row, free actors' rows: 43 values for the pushing scene) is not touched while the pairs are
// walked, and the link poses the solve needs afterwards are the frames scene_frames() has just written to the sample's LDS
// rows: the owner's leader lane parks the state in the sample's row before the walk, every lane reads it back after it (same
// wavefront, LDS operations in order, bit-identical values), and the poses are re-read from the frames instead of being kept.
template <class T>
MPPI_HD void state_park(const SceneState<T> &s, int n_free, const LMem &L, bool leader) {
    if (!leader) return;
    constexpr int NB = T::NB;
    int o = L.park;
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        L[o + i] = s.q[i];
        L[o + NB + i] = s.qd[i];
    });
    o += 2 * NB;
    for (int j = 0; j < 13; j++) L[o + j] = s.base[j];
    o += 13;
    for (int f = 0; f < kFreeSlots; f++)
        if (f < n_free)
            for (int j = 0; j < 13; j++) L[o + 13 * f + j] = s.fr[f][j];
}
template <class T>
MPPI_HD void state_unpark(SceneState<T> &s, int n_free, const LMem &L) {
    constexpr int NB = T::NB;
    int o = L.park;
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        s.q[i] = L[o + i];
        s.qd[i] = L[o + NB + i];
    });
    o += 2 * NB;
    for (int j = 0; j < 13; j++) s.base[j] = L[o + j];
    o += 13;
    for (int f = 0; f < kFreeSlots; f++)
        if (f < n_free)
            for (int j = 0; j < 13; j++) s.fr[f][j] = L[o + 13 * f + j];
}
// link poses and base velocity back from the frames of this substep (what scene_frames stored)
template <class T, class M>
MPPI_HD void pose_from_frames(M &m, const LMem &L, Pose<T> &P, BaseSV<T> &vbase) {
    constexpr int NB = T::NB;
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        const int o = i * 18;
        for (int j = 0; j < 9; j++) P.R[i].a[j] = L[o + j];
        P.p[i] = {L[o + 9], L[o + 10], L[o + 11]};
    });
    SV vb;
    frame_load(L, NB, P.Rb, P.pb, vb);   // (the helper-wavefront kernel: one base - several bases run on the one-lane kernels)
    vbase.r[0] = m.floating ? vb : SV{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
}

// One simulator step of a contact scene (dt = substeps * h).  `root` carries the static actors.
template <class T, int SPLIT = kSplitNone, class M = CModel>
MPPI_HD void step_scene(M &m0, const float *root, SceneState<T> &s, const float *target, const LMem &L, Split split = Split{0, 1}) {
    constexpr int NB = T::NB;
    M *mp = &m0;
    // position mode (reference isaacgym_wrapper.py:571-572): apply_robot_cmd overwrites the DOF state with the command
    if (m0.drive_mode == kDrivePosition)
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA { s.q[ic] = target[ic]; s.qd[ic] = 0.f; });
    for (int sub = 0; sub < m0.substeps; sub++) {
        M &m = *launder(mp);
        // (position drive: the spring at the end-of-substep position = damping kd + h kp, mppi_device.hpp step)
        const bool posmode = m.drive_mode == kDrivePosition;
        const float h = m.h, kp = posmode ? m.kp : 0.f, kd = m.kd + h * kp;
        Pose<T> P;
        BaseSV<T> vbase;
        scene_frames<T>(m, root, s, P, vbase, L);
        MPPI_SEC(0);
#if defined(MPPI_NO_PARK)
        constexpr bool kPark = false;
#else
        constexpr bool kPark = SPLIT == kSplitOctPair;
#endif
        constexpr int kParkTarget = 2 * NB + 13 + 13 * kFreeSlots;  // (behind the state, see scene_park_floats)
        float tgt[NB ? NB : 1];
        if constexpr (kPark) {
            state_park<T>(s, m.n_free, L, split.sub == 0);
            if (sub == 0 && split.sub == 0)  // the drive targets of this step: the caller's registers are free from here on
                static_for<0, NB>([&](auto ic) MPPI_LAMBDA { L[L.park + kParkTarget + ic] = target[ic]; });
        }
        contact_forces<T, SPLIT>(m, root, L, s.acc_dirty, s.cf_dirty, split);
        if constexpr (kPark) {
            M &m2 = *launder(mp);
            state_unpark<T>(s, m2.n_free, L);
            pose_from_frames<T>(m2, L, P, vbase);
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA { tgt[ic] = L[L.park + kParkTarget + ic]; });
        } else {
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA { tgt[ic] = target[ic]; });
        }
        float tau[NB ? NB : 1], kdh[NB ? NB : 1], qdd[NB ? NB : 1], ff[NB ? NB : 1], vs[NB ? NB : 1];
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            ff[i] = m.drive_mode == kDriveEffort ? tgt[i] : (posmode ? kp * (tgt[i] - s.q[i]) : 0.f);
            vs[i] = m.drive_mode == kDriveVelocity ? tgt[i] : 0.f;
            tau[i] = ff[i] + kd * (vs[i] - s.qd[i]);
            kdh[i] = kd * h;
        });
        BaseSV<T> abase;
        aba_scene<T>(m, P, vbase, s.qd, tau, kdh, L, qdd, abase);
        MPPI_SEC(5);
        bool any = false;
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            float lim = m.b[i].k1.effort;
            float tt = ff[i] + kd * (vs[i] - s.qd[i] - h * qdd[i]);
            if (lim > 0.f && fabsf(tt) > lim) {
                any = true;
                tau[i] = tt > 0.f ? lim : -lim;
                kdh[i] = 0.f;
            }
        });
        if (any) aba_scene<T>(*launder(mp), P, vbase, s.qd, tau, kdh, L, qdd, abase);
        MPPI_SEC(6);
        float dqd[NB ? NB : 1];   // (rate changes of the substep: step_free_bodies)
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            const BodyK1 b = load_block<BodyK1>(m.b[i].k1);
            float v = s.qd[i] + h * qdd[i];
            if (b.vmax > 0.f) v = fminf(fmaxf(v, -b.vmax), b.vmax);
            float x = s.q[i] + h * v;
            const float lo = m.b[i].k0.lower, hi = m.b[i].k0.upper;
            if (lo > -INFINITY || hi < INFINITY) joint_limit(s.q[i], x, v, lo, hi, 1.f / h);
            s.q[i] = x;
            dqd[i] = v - s.qd[i];
            s.qd[i] = v;
        });
        if (m.floating)
            static_for<0, T::NBASE>([&](auto rc) MPPI_LAMBDA { root_integrate(s.template base_row<(int)rc>(), abase.r[rc], h); });
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (SPLIT == kSplitOctPair) {
            // the helper wavefront has solved the free actors while this one solved the robot (helper_free_bodies)
            MPPI_BARRIER(4);
            for (int f = 0; f < kFreeSlots; f++)
                if (f < m.n_free) {
                    const int o = L.xch + 2 + 6 * f;
                    root_integrate(s.fr[f], SV{{L[o], L[o + 1], L[o + 2]}, {L[o + 3], L[o + 4], L[o + 5]}}, h);   // (the limit: helper_free_bodies)
                }
        } else
#endif
        step_free_bodies<T>(m, s, L, h, dqd);
        MPPI_SEC(7);
    }
}

// ---- scene rollouts -----------------------------------------------------------------------------
template <class T, class M>
MPPI_HD void scene_init(M &m, const float *dof0, const float *root, SceneState<T> &s, int g, const LMem &L) {
    scene_randomise<T>(m, g, L);
    static_for<0, T::NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        s.q[i] = dof0[2 * i];
        s.qd[i] = dof0[2 * i + 1];
    });
    static_for<0, T::NBASE>([&](auto rc) MPPI_LAMBDA {
        constexpr int r = rc;
        for (int j = 0; j < 13; j++) s.template base_row<r>()[j] = root[13 * base_actor<r>(m) + j];
    });
    for (int f = 0; f < kFreeSlots; f++)
        for (int j = 0; j < 13; j++) s.fr[f][j] = f < m.n_free ? root[13 * m.fr[f].actor + j] : 0.f;
}

// yaw of an xyzw quaternion (reference mppiisaac/utils/conversions.py:4-11)
MPPI_HD float quat_yaw(const float *q) {
    return atan2f(2.f * (q[3] * q[2] + q[0] * q[1]), q[3] * q[3] + q[0] * q[0] - q[1] * q[1] - q[2] * q[2]);
}

// Stage cost of a contact scene.  BOXER_PUSH restates examples/boxer_push/planner.py:26-67:
// link[0] = robot link (ee_link), actor[0] = block, actor[1] = goal, link[1], link[2] = rigid bodies of the
// two obstacles; w = {robot_to_block, block_to_goal, block_to_goal_ort, push_align, velocity, collision, goal_yaw}.
// core: R, r = world pose of link[0] (callers obtain it from their own kinematics)
template <class T, class M>
MPPI_HD float stage_cost_scene_link(M &m, CCost &c, const float *root, const SceneState<T> &s, const LMem &L, const M3 &R, V3 r) {
    using Lay = SceneLayout<T>;
    if (c.kind == kCostBoxerPush) {
        // block = free actor (looked up by actor id), goal = static actor
        float bx = 0.f, by = 0.f, bvx = 0.f, bvy = 0.f, byaw = 0.f;
        for (int f = 0; f < kFreeSlots; f++)
            if (f < m.n_free && m.fr[f].actor == c.actor[0]) {
                bx = s.fr[f][0]; by = s.fr[f][1]; bvx = s.fr[f][7]; bvy = s.fr[f][8];
                byaw = quat_yaw(s.fr[f] + 3);
            }
        const float gx = root[13 * c.actor[1]], gy = root[13 * c.actor[1] + 1];
        const float rbx = r.x - bx, rby = r.y - by, bgx = gx - bx, bgy = gy - by;
        const float d_rb = fsqrt(rbx * rbx + rby * rby), d_bg = fsqrt(bgx * bgx + bgy * bgy);
        const float ort = fabsf(byaw - c.w[6]);
        const float align = (rbx * bgx + rby * bgy) * frcp(d_rb * d_bg) + 1.f;
        const int o1 = Lay::kCf + 3 * c.link[1], o2 = Lay::kCf + 3 * c.link[2];
        const float coll = fabsf(L[o1]) + fabsf(L[o1 + 1]) + fabsf(L[o2]) + fabsf(L[o2 + 1]);
        const float vel = fsqrt(bvx * bvx + bvy * bvy);
        return c.w[0] * d_rb + c.w[1] * d_bg + c.w[2] * ort + c.w[3] * align + c.w[4] * vel + c.w[5] * coll;
    }
    // PANDA_PICK, examples/panda_pick/planner.py:24-53: link[0] = panda_ee, link[1] = table rigid body, actor[0] = block,
    // actor[1] = goal; w = {robot_to_block, block_to_goal, collision, robot_ori}
    V3 b = {0.f, 0.f, 0.f};
    for (int f = 0; f < kFreeSlots; f++)
        if (f < m.n_free && m.fr[f].actor == c.actor[0]) b = loadv(s.fr[f]);
    const V3 g = loadv(root + 13 * c.actor[1]);
    const V3 drb = r - b, dbg = b - g;
    const int ot = Lay::kCf + 3 * c.link[1];
    const float forces = fabsf(L[ot]) + fabsf(L[ot + 1]) + fabsf(L[ot + 2]);
    const float a0 = atan2f(R.a[7], -R.a[8]), a1 = asinf(clamp1(R.a[6]));  // see PANDA_REACH
    return c.w[0] * fsqrt(dot(drb, drb)) + c.w[1] * fsqrt(dot(dbg, dbg)) + c.w[2] * forces + c.w[3] * fsqrt(a0 * a0 + a1 * a1);
}

// env of a contact scene as a cost program sees it: the robot's root row is the sample's own base state, free actors their
// own rows, everything else the static x0 rows; net contact forces from the sample's LDS rows
template <class T, class M>
struct SceneEnv {
    M &m;
    const float *root;
    const SceneState<T> &s;
    const LMem &L;
    MPPI_HD float at(int actor, int j) const {
        float v = root[13 * actor + j];
        if (actor == m.robot_actor && m.floating) v = s.base[j];
        if constexpr (T::NBASE > 1)
            static_for<1, T::NBASE>([&](auto rc) MPPI_LAMBDA { if (actor == m.xbase_actor[rc - 1]) v = s.xbase[rc - 1][j]; });
        for (int f = 0; f < kFreeSlots; f++)
            if (f < m.n_free && m.fr[f].actor == actor) v = s.fr[f][j];
        return v;
    }
    MPPI_HD V3 vec(int actor, int off) const {
        V3 o = loadv(root + 13 * actor + off);
        if (actor == m.robot_actor && m.floating) o = loadv(s.base + off);
        if constexpr (T::NBASE > 1)
            static_for<1, T::NBASE>([&](auto rc) MPPI_LAMBDA { if (actor == m.xbase_actor[rc - 1]) o = loadv(s.xbase[rc - 1] + off); });
        for (int f = 0; f < kFreeSlots; f++)
            if (f < m.n_free && m.fr[f].actor == actor) o = loadv(s.fr[f] + off);
        return o;
    }
    MPPI_HD void quat(int actor, float *qq) const {
        for (int j = 0; j < 4; j++) qq[j] = root[13 * actor + 3 + j];
        if (actor == m.robot_actor && m.floating)
            for (int j = 0; j < 4; j++) qq[j] = s.base[3 + j];
        if constexpr (T::NBASE > 1)
            static_for<1, T::NBASE>([&](auto rc) MPPI_LAMBDA {
                if (actor == m.xbase_actor[rc - 1])
                    for (int j = 0; j < 4; j++) qq[j] = s.xbase[rc - 1][3 + j];
            });
        for (int f = 0; f < kFreeSlots; f++)
            if (f < m.n_free && m.fr[f].actor == actor)
                for (int j = 0; j < 4; j++) qq[j] = s.fr[f][3 + j];
    }
    MPPI_HD float cf(int rb, int j) const { return L[SceneLayout<T>::kCf + 3 * rb + j]; }
    MPPI_HD V3 constant_point(float x, float y, float z) const { return V3{x - L.ox, y - L.oy, z}; }  // (rollout coordinates, see root_relative)
};

// base poses of the sample's own state into P (what forward_kinematics_base starts from)
template <class T>
MPPI_HD void pose_bases(const SceneState<T> &s, Pose<T> &P) {
    static_for<0, T::NBASE>([&](auto rc) MPPI_LAMBDA {
        constexpr int r = rc;
        const float *bs = s.template base_row<r>();
        P.template base_p<r>() = loadv(bs);
        P.template base_R<r>() = quat_to_R(bs + 3);
    });
}
template <class T, class M>
MPPI_HD float stage_cost_scene(M &m, CCost &c, const float *root, const SceneState<T> &s, const LMem &L) {
    Pose<T> P;
    pose_bases<T>(s, P);
    if (c.kind == kCostProgram) {
        forward_kinematics_base<T>(m, s.q, P);
        return program_cost<T>(m, c, s.q, s.qd, P, SceneEnv<T, M>{m, root, s, L});
    }
    if (c.kind == kCostBoxerPush || c.kind == kCostPandaPick) {
        forward_kinematics_base<T>(m, s.q, P);
        M3 R;
        V3 r;
        link_pose<T>(m, P, c.link[0], R, r);
        return stage_cost_scene_link<T>(m, c, root, s, L, R, r);
    }
    // reach costs: the link pose comes from the sample's own (possibly floating) base
    if (c.kind == kCostPandaReach) forward_kinematics_base<T>(m, s.q, P);
    return stage_cost_pose<T>(m, c, root, s.q, P);
}

// quad layout of the robot's kinematics / articulated-body solve (mppi_scene_quad.hpp; used by the kernels whose lanes
// share a sample)
template <class T, int SPLIT, class M, class MR>
MPPI_HD void step_scene_quad(M &m0, MR &mr0, const float *root, SceneState<T> &s, const float *target, const LMem &L, Split split);
// stage cost + rollout-visualisation point of the state after a step, both from ONE quad-layout kinematics pass
template <class T, class M, class MR>
MPPI_HD float step_tail_scene_quad(M &m, MR &mr, CCfg &cfg, CCost &c, const float *root, const SceneState<T> &s, const LMem &L, float *viz, int t, int k,
                                   bool leader);

#if defined(__HIP_DEVICE_COMPILE__)
template <class T, int SPLIT, class M, class MR>
__device__ __forceinline__ void step_scene_oct(M &m0, MR &mr0, const float *root, SceneState<T> &s, const float *target, const LMem &L, Split split);
#endif
// One step of a contact scene in the layout the kernel runs in.  The quad layout of the robot algebra pays from a handful
// of bodies on (measured: gripper arm, 9 bodies, 5.11 -> 4.43 ms; boxer, 2 wheels on a floating base, 1.64 -> 1.87 ms):
// short trees keep the replicated one-lane algebra.
template <class T, int SPLIT, class M, class MR>
MPPI_HD void step_scene_any(M &m, MR &mr, const float *root, SceneState<T> &s, const float *target, const LMem &L, Split split) {
    if constexpr (SPLIT == kSplitNone || T::NB <= 4) step_scene<T, SPLIT>(m, root, s, target, L, split);
    else {
#if defined(__HIP_DEVICE_COMPILE__)
        // fixed-base trees in the octet kernel: the solve with the angular / linear halves of every spatial quantity in the two
        // quads of the sample (mppi_scene_oct.hpp) instead of the quad-layout solve computed by both quads alike - a kernel
        // instantiation of its own (both solves in one kernel cost the register allocation of the larger plus spills)
        if constexpr (SPLIT == kSplitOctSolve) step_scene_oct<T, SPLIT>(m, mr, root, s, target, L, split);
        else
#endif
            step_scene_quad<T, SPLIT>(m, mr, root, s, target, L, split);
    }
}

// mr0: view of the same model for the robot algebra of the quad path (an LDS copy of the model prefix in the kernel)
// DUMP: the env state after every step goes to `traj` as well, in the sample-minor layout of the simulator's state arrays with
// H*K columns (column t*K + k): q [NB], qd [NB], base [13], free actors [kFreeSlots*13], net contact forces [3*n_rb] - the generic
// Objective mode evaluates Python costs on the materialised trajectory (mppi_rollout_trajectory).
template <class T, int SPLIT = kSplitNone, bool DUMP = false, class M = CModel, class MR = CModel>
MPPI_HD float rollout_scene(M &m0, MR &mr0, CCfg &cfg0, CCost &cost0, const float *dof0, const float *root, const float *U, const float *eps,
                            const float *prior, float *du, float *viz, int k, const LMem &L, Split split = Split{0, 1}, float *traj = nullptr) {
    const bool leader = split.sub == 0;  // lanes sharing a sample hold identical values: one of them writes
    constexpr int NB = T::NB;
    const int K = cfg0.K, nu = cfg0.nu, H = cfg0.H;
    const int g = cfg0.k_offset + k;
    const bool is_null = cfg0.sample_null_action && g == cfg0.k_total - 1;
    const bool is_prior = cfg0.use_priors && prior != nullptr && g == cfg0.k_total - 2;
    SceneState<T> s;
    scene_init<T>(m0, dof0, root, s, g, L);
    if constexpr (SPLIT != kSplitNone) shape_cache_update<T, split_octet(SPLIT)>(m0, root, L, SPLIT == kSplitEmulate ? Split{0, 1} : split, true);
    float target[NB ? NB : 1], u[kMaxNu];
    float S = 0.f, ctrl = 0.f, disc = 1.f;
    M *mp = &m0;
    CCfg *cp = &cfg0;
    CCost *kp = &cost0;
    MPPI_SEC(10);
    for (int t = 0; t < H; t++) {
        CCfg &cfg = *launder(cp);
        ctrl += sample_controls<(NB < kMaxNu ? (NB ? NB : 1) : kMaxNu)>(cfg, U, eps, prior, t, k, is_null, is_prior, leader, du, u);
        if constexpr (SPLIT == kSplitNone) cmd_map<T>(*launder(mp), u, target);
        else cmd_map<T>(mr0, u, target);  // (the kernel's LDS copy of the robot part: no scalar-cache round trips)
        MPPI_SEC(8);
        constexpr bool kParkSums = SPLIT == kSplitOctPair;  // (the helper-wavefront kernel: everything that idles during a step is parked)
        constexpr int kParkSumsAt = 2 * NB + 13 + 13 * kFreeSlots + (NB ? NB : 1);
        if constexpr (kParkSums)
            if (leader) { L[L.park + kParkSumsAt] = S; L[L.park + kParkSumsAt + 1] = ctrl; L[L.park + kParkSumsAt + 2] = disc; }
        step_scene_any<T, SPLIT>(*mp, mr0, root, s, target, L, split);
        if constexpr (kParkSums) { S = L[L.park + kParkSumsAt]; ctrl = L[L.park + kParkSumsAt + 1]; disc = L[L.park + kParkSumsAt + 2]; }
        if constexpr (SPLIT == kSplitNone || T::NB <= 4) {
            S += disc * stage_cost_scene<T>(*launder(mp), *launder(kp), root, s, L);
            if (cfg.want_rollouts && viz != nullptr && leader) {
                M &m = *launder(mp);
                Pose<T> P;
                pose_bases<T>(s, P);
                forward_kinematics_base<T>(m, s.q, P);
                M3 R;
                V3 p;
                link_pose<T>(m, P, cfg.viz_link, R, p);
                // (32-bit element indices, 3 H K < 2^31 by pack_config: a 64-bit per-lane index is two registers for the whole rollout)
                viz[(unsigned)(t * 3 + 0) * (unsigned)K + (unsigned)k] = p.x + L.ox;
                viz[(unsigned)(t * 3 + 1) * (unsigned)K + (unsigned)k] = p.y + L.oy;
                viz[(unsigned)(t * 3 + 2) * (unsigned)K + (unsigned)k] = p.z;
            }
        } else {
            S += disc * step_tail_scene_quad<T>(*launder(mp), mr0, cfg, *launder(kp), root, s, L, viz, t, k, leader);
        }
        if constexpr (DUMP) {
            if (leader) {
                const size_t HK = (size_t)H * K, col = (size_t)t * K + k;
                float *o = traj + col;
                static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
                    constexpr int i = ic;
                    o[(size_t)i * HK] = s.q[i];
                    o[(size_t)(NB + i) * HK] = s.qd[i];
                });
                o += (size_t)2 * NB * HK;
                for (int j = 0; j < 13; j++) o[(size_t)j * HK] = s.base[j] + (j == 0 ? L.ox : (j == 1 ? L.oy : 0.f));
                o += (size_t)13 * HK;
                for (int f = 0; f < kFreeSlots; f++)
                    for (int j = 0; j < 13; j++) o[(size_t)(f * 13 + j) * HK] = s.fr[f][j] + (j == 0 ? L.ox : (j == 1 ? L.oy : 0.f));
                o += (size_t)13 * kFreeSlots * HK;
                const int n_cf = 3 * launder(mp)->n_rb;
                for (int j = 0; j < n_cf; j++) o[(size_t)j * HK] = L[SceneLayout<T>::kCf + j];
            }
        }
        MPPI_SEC(9);
        disc *= cfg.gamma;
    }
    return S + ctrl;
}

// reference-layout rows of ONE env of a contact scene: root [A][13], rigid bodies [n_rb][13], contact forces [n_rb][3]
template <class T, class M>
MPPI_HD void scene_materialise(M &m, const float *root, const SceneState<T> &s, const float *cf_in /* [n_rb*3] or null */,
                               float *root_out, float *rb, float *cf) {
    constexpr int NB = T::NB;
    // root rows: static actors from x0, robot and free actors from the env state
    if (root_out != nullptr) {
        for (int j = 0; j < 13 * m.n_actors; j++) root_out[j] = root[j];
        static_for<0, T::NBASE>([&](auto rc) MPPI_LAMBDA {
            constexpr int r = rc;
            for (int j = 0; j < 13; j++) root_out[13 * base_actor<r>(m) + j] = s.template base_row<r>()[j];
        });
        for (int f = 0; f < kFreeSlots; f++)
            if (f < m.n_free)
                for (int j = 0; j < 13; j++) root_out[13 * m.fr[f].actor + j] = s.fr[f][j];
    }
    if (rb != nullptr) {
        Pose<T> P;
        pose_bases<T>(s, P);
        forward_kinematics_base<T>(m, s.q, P);
        BaseSV<T> vbase;
        static_for<0, T::NBASE>([&](auto rc) MPPI_LAMBDA {
            constexpr int r = rc;
            const float *bs = s.template base_row<r>();
            V3 wb = loadv(bs + 10), vb = loadv(bs + 7);
            vbase.r[r] = m.floating ? SV{wb, vb - cross(wb, P.template base_p<r>())} : SV{{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        });
        SV v[NB ? NB : 1];
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            constexpr int par = T::par[i];
            SV S = joint_subspace<T, i>(m, P);
            SV sj = {s.qd[i] * S.a, s.qd[i] * S.l};
            if constexpr (par < 0) v[i] = vbase.r[base_of_parent(par)] + sj;
            else v[i] = v[par < 0 ? 0 : par] + sj;
        });
        for (int a = 0; a < m.n_actors; a++) {
            if (a == m.robot_actor) continue;
            if (T::NBASE > 1 && m.actor_first_rb[a] >= m.robot_first_rb && m.actor_first_rb[a] < m.robot_first_rb + m.nl) continue;  // (a further robot of the forest: its links' rows are written below)
            float *o = rb + 13 * m.actor_first_rb[a];
            for (int j = 0; j < 13; j++) o[j] = root[13 * a + j];
        }
        for (int f = 0; f < kFreeSlots; f++)
            if (f < m.n_free) {
                float *o = rb + 13 * m.fr[f].rb;
                for (int j = 0; j < 13; j++) o[j] = s.fr[f][j];
            }
        for (int l = 0; l < m.nl; l++) {
            M3 R;
            V3 p;
            link_pose<T>(m, P, l, R, p);
            SV vl = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
            static_for<0, T::NBASE>([&](auto rc) MPPI_LAMBDA {
                constexpr int r = rc;
                const float w0 = (T::NBASE > 1 ? m.l[l].body == -1 - r : m.l[l].body < 0) ? 1.f : 0.f;
                vl = {vl.a + w0 * vbase.r[r].a, vl.l + w0 * vbase.r[r].l};
            });
            static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
                constexpr int i = ic;
                const float w = m.l[l].body == i ? 1.f : 0.f;
                vl = {vl.a + w * v[i].a, vl.l + w * v[i].l};
            });
            V3 lv = vl.l + cross(vl.a, p);
            float *o = rb + 13 * (m.robot_first_rb + l);
            o[0] = p.x; o[1] = p.y; o[2] = p.z;
            R_to_quat(R, o + 3);
            o[7] = lv.x; o[8] = lv.y; o[9] = lv.z;
            o[10] = vl.a.x; o[11] = vl.a.y; o[12] = vl.a.z;
        }
    }
    if (cf != nullptr)
        for (int j = 0; j < 3 * m.n_rb; j++) cf[j] = cf_in != nullptr ? cf_in[j] : 0.f;
}

}  // namespace mppi
