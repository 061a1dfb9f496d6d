// mppi_device.hpp - per-sample device code of the MPPI rollout (gfx950).
//
// One lane integrates one sample: the whole articulated-body working set (joint subspaces,
// articulated inertias, bias forces) lives in VGPRs - with <=2 waves per SIMD the lane has the
// full 512-entry unified register file - and the kinematic tree is a COMPILE-TIME template
// parameter, so every per-body array index is static and nothing is spilled or indexed through
// scratch.  All spatial quantities are expressed in WORLD axes about the WORLD origin, which
// makes the inertia/force propagation of the articulated-body algorithm a plain addition
// (no 6x6 congruence transforms in the hot loop).
//
// The model handed to this code is "z-framed": the host re-frames every body so that its joint
// axis is the local z axis (csrc/mppi_pack.hpp: pack_model), so a revolute joint
// costs one sincos and 12 multiplies.
//
// This header is plain C++17: hipcc compiles it for the GPU kernels (mppi_hip.hip); the
// test-only host harness (tests/hostemu) compiles the very same functions with g++ so the
// arithmetic can be checked against the oracle without a GPU.  It is NOT a product CPU path.
//
// Reference behaviour restated here (paths in the reference tree):
//   apply_robot_cmd / _ik   mppiisaac/planner/isaacgym_wrapper.py:510-572
//   step (dt, substeps)     mppiisaac/planner/isaacgym_wrapper.py:639-645, conf/isaacgym/*.yaml
//   drive gains             mppiisaac/planner/isaacgym_wrapper.py:491-507
//   stage costs             examples/panda/planner.py:22-40, benchmarks/point_robot/.../mppi_planner_wrapper.py:17-35
#pragma once
#include <math.h>
#include <stdint.h>
#include <initializer_list>

#if defined(__HIPCC__)
#define MPPI_HD __host__ __device__ __forceinline__
#else
#define MPPI_HD inline
#endif

namespace mppi {

constexpr int kMaxBodies = 12;
constexpr int kMaxLinks = 24;
constexpr int kMaxActors = 12;
constexpr int kMaxNu = 12;
constexpr int kMaxShapes = 64;
constexpr int kMaxPairs = 128;
constexpr int kPairKernelMaxPairs = 64;  // the helper-wavefront scene kernel: two mask words (mppi_scene.hpp, pair groups)
constexpr int kMaxFree = 4;        // free actors a MODEL may hold (MPPI_MAX_FREE)
// free-actor slots the contact-scene KERNELS of this build carry (state rows, frames, LDS rows are sized by it): 2 in the shipped
// library - every example scene of the reference has at most two free actors, and two more slots cost the register-bound scene
// kernels 26 state values and two frames per sample -; a scene with 3 or 4 free actors gets its kernels from an on-demand
// build with -DMPPI_FREE_SLOTS=4 (mppi_hip.hip: jit_topology)
#ifndef MPPI_FREE_SLOTS
#define MPPI_FREE_SLOTS 2
#endif
constexpr int kFreeSlots = MPPI_FREE_SLOTS;
static_assert(kFreeSlots >= 1 && kFreeSlots <= kMaxFree, "MPPI_FREE_SLOTS out of range");
constexpr int kMaxExtraBases = 3;  // MPPI_MAX_EXTRA_BASES (checked in mppi_pack.hpp)

// ---- device-side model (fp32, z-framed) ------------------------------------------------
struct CmdBlock {
    float v[16];
};
// Three 64-byte blocks, each fetched with ONE s_load_dwordx16 where it is used (load_block below):
// block 0 = kinematics, block 1 = inertia + limits, block 2 = command map row.
struct BodyK0 {
    // joint frame in the parent body's frame (child->parent) at q = 0 as ONE row-major 3x4 block [Rt | pt]: the quad layout's
    // kinematics multiply register PAIRS (v_pk_fma_f32), and (Rt[r][0], Rt[r][1]) and (Rt[r][2], pt[r]) are such pairs as the
    // block arrives from LDS (mppi_quad.hpp quad_fk)
    float T[12];
    int jtype, parent;
    float lower, upper;  // joint range (-inf / +inf: none)
    MPPI_HD float rt(int j) const { return T[4 * (j / 3) + j % 3]; }  // Rt[j], row-major 3x3
    MPPI_HD float pt(int r) const { return T[4 * r + 3]; }
};
struct BodyK1 {
    // first moment hb = m * com and inertia about the COM Ic (xx xy xz yy yz zz), body axes, laid out as the constant PAIRS of
    // the quad layout's packed products (mppi_quad.hpp quad_aba):  (h, Tr0) = c0 (hb0, Ic0) + c1 (hb1, Ic1) + c2 (hb2, Ic2) and
    // (Tr1, Tr2) = c0 (Ic1, Ic2) + c1 (Ic3, Ic4) + c2 (Ic4, Ic5)  for the columns c of R;  hb() / Ic() read single entries
    float hI[6];  // hb0 Ic0 | hb1 Ic1 | hb2 Ic2
    float II[6];  // Ic1 Ic2 | Ic3 Ic4 | Ic4 Ic5
    float m, invm;  // mass, 1/m (0 for massless bodies)
    float effort, vmax;  // +inf: no limit
    MPPI_HD float hb(int j) const { return hI[2 * j]; }
    MPPI_HD float Ic(int k) const { return k < 3 ? hI[2 * k + 1] : (k == 3 ? II[2] : (k == 4 ? II[3] : II[5])); }
    MPPI_HD void set(const float *hb3, const float *Ic6) {
        for (int j = 0; j < 3; j++) { hI[2 * j] = hb3[j]; hI[2 * j + 1] = Ic6[j]; }
        II[0] = Ic6[1]; II[1] = Ic6[2]; II[2] = Ic6[3]; II[3] = Ic6[4]; II[4] = Ic6[4]; II[5] = Ic6[5];
    }
};
struct alignas(64) DevBody {
    BodyK0 k0;
    BodyK1 k1;
    CmdBlock cmd;  // dense row of the command map: target_i = sum_c cmd[c] * u[c]  (zero beyond nu)
};
static_assert(sizeof(BodyK0) == 64 && sizeof(BodyK1) == 64 && sizeof(DevBody) == 192, "DevBody block layout");
struct DevLink {
    int body, pad[3];
    float R[9];
    float p[3];
};
// Contact scene.  A "frame" is anything a shape can be welded to: moving body i (0..nb-1), the robot
// base (nb) and free actor f (nb+1+f) are DYNAMIC frames whose pose/velocity live in per-sample
// memory; everything else is static and posed by the root state of `src_actor`.
struct DevShape {
    int ent;        // dynamic frame index, -1 = static
    int src_actor;  // static shapes: actor whose root pose carries the shape
    int type, rb;   // MPPI_SHAPE_*, rigid-body row for net_contact_force
    float half[3];  // box half extents | radius in half[0]
    float mu;
    float R[9], p[3];
};
// One candidate pair = two 64-byte blocks.  The broad phase needs block g only (one s_load_dwordx16: shape ids for the
// pose cache, types, half extents - no dependent loads of the two shape records); block c (contact law, noise lookup)
// is fetched for the pairs that survive it.
struct PairGeom {
    int a, b;        // shapes; b = -1: ground plane
    int mode;        // 0: both dynamic (explicit penalty), 1: a dynamic / b static (implicit), 2: b dynamic / a static,
                     // 3 / 4: both dynamic, b / a a LIGHT free actor held implicitly by the robot link a / b (mppi_scene.hpp)
    int rnd;         // a noisy actor takes part: size / friction / mass scale are per sample
    int typeA, typeB, entA, entB;  // MPPI_SHAPE_* (typeB = -1: ground), dynamic frame (-1: static)
    float hA[3], hB[3];            // nominal half extents | radius in [0]
    int rbA, rbB;                  // rigid-body rows for net_contact_force
};
struct PairGain {
    float mu, k, cn, ct;  // combined friction, stiffness and damping per contact point
    float kh;             // k * h (implicit spring term of static contacts)
    float ma, mb, mub;    // nominal masses of the reacting actors (-1: static) and the ground friction for b = -1
    int actorA, actorB;   // actors of the two shapes (rows of the per-sample noise draws)
    int robotA, robotB;   // the shape belongs to the robot (never noisy; keeps its own per-shape friction)
    float muA, muB;       // per-shape friction
    float inv_d0;         // 1 / contact_ramp_depth (0: no ramp)
    float npts;           // nominal patch size (4: box against box / ground, else 1); mode-0 pairs carry per-point gains for npts = 1
};
struct DevPair {
    PairGeom g;
    PairGain c;
};
struct DevFree {
    int actor, rb, gravity, type;
    float m, Ic[3];  // box / sphere principal inertia about the centre
    float size[3], pad;
};
struct DevModel {
    int nb, nl, n_actors, robot_actor, n_rb, robot_first_rb, drive_mode, substeps, gravity_on, nu, floating, n_free;
    float kd, h, g[3], base_m;
    int cmd_identity;  // the command map is the identity (nu == nb, one command per body, unit gain): target = u
    int all_revolute;  // every joint is revolute: the quad rollout runs its compile-time specialisation
    float base_hb[3], base_Ic[6];
    float kp;  // position drive stiffness (kDrivePosition; 0 otherwise)
    // round 6: candidate pairs between a robot link and a free actor at least MPPI_LIGHT_BODY_RATIO times lighter than the robot
    // (PairGeom::mode 3 / 4: implicit on both bodies, mppi_scene.hpp "light bodies"), and the free slots (bit f) such an actor sits in
    int n_light_pairs;
    unsigned light_free;
    int actor_first_rb[kMaxActors];
    int n_shapes, n_pairs, rnd_seed, n_rnd;
    int rnd_slot[kMaxActors];    // LDS slot of a noisy actor's per-sample draws (-1: nominal)
    float noise[kMaxActors][5];  // per actor: sigma_size xyz, mass percentage, friction percentage
    float actor_mu[kMaxActors], actor_mass[kMaxActors];
    DevBody b[kMaxBodies];
    DevLink l[kMaxLinks];
    DevFree fr[kMaxFree];
    DevShape sh[kMaxShapes];
    DevPair pr[kMaxPairs];
    // the further moving bases of the env (ABI 7; behind everything else: no offset of the single-robot models moves): base r > 0
    // is xbase_*[r - 1], base 0 is robot_actor / base_m / base_hb / base_Ic above.  Read by the one-lane scene kernels only.
    int n_bases;
    int xbase_actor[kMaxExtraBases];
    float xbase_m[kMaxExtraBases], xbase_hb[kMaxExtraBases][3], xbase_Ic[kMaxExtraBases][6];
    // pair groups (round 5; shared-lane scene kernels): all candidate pairs between the ROBOT's shapes and one shape of another
    // actor form a group; when that shape is further from the robot's anchor shape than the robot can reach - in every sample of
    // the wavefront - none of the group's pairs is visited (contact_forces: a culled pair still costs its record, two shape poses
    // and the broad-phase arithmetic, ~350 issue slots; the pushing scene has 15 robot-block / robot-obstacle pairs since the
    // wheels and casters meet boxes).  Conservative like the broad phase: a skipped pair is one the broad phase would have culled.
    int n_groups;
    int pair_normal;  // two dynamic boxes: one normal per pair from the separating-axis test (box_pair_sat); 0: the per-point law of ABI <= 7
    struct Group {
        int anchor, other;          // shapes whose cached centres are compared: a robot shape welded to base 0, the other actor's shape
        unsigned mask_lo, mask_hi;  // the group's pairs (0-31, 32-63)
        float reach2;               // (robot reach about the anchor's centre + bounding radius of the other shape + margins)^2
        unsigned mask_2, mask_3;    // ... pairs 64-95, 96-127
        float pad;
    } grp[8];
};
constexpr int kMaxGroups = 8;
struct CtrlBlock {  // one 64-byte block per quantity: fetched with a single s_load_dwordx16
    float v[16];
};
struct alignas(64) DevCfg {
    int K, H, nu, k_offset, k_total, sample_null_action, use_priors, noise_abs_cost, want_rollouts, viz_link;
    float *action_mirror;  // host-mapped pinned copy of the action (written by the update kernels: no D2H copy op), or null
    float lambda, inv_lambda, gamma, u_init;
    unsigned *seq_dev, *seq_host;  // update counter (device) and its host-mapped mirror, published after the action
    CtrlBlock u_min, u_max, inv_sigma;  // per control dimension, zero beyond nu; inv_sigma = 1 / noise_sigma[c][c]
};

// one term of a cost program (include/mppi_hip.h mppi_term_t after pack_cost): operand kinds are resolved for the device -
// a rigid body is either a robot link (kSrcLink, robot-local index) or the single body of a box / sphere actor (kSrcActor)
enum { kSrcNone = 0, kSrcLink = 1, kSrcActor = 2, kSrcDofXY = 3, kSrcConst = 4 };
enum { kOpDist = 1, kOpTilt = 2, kOpYawAbs = 3, kOpAlign = 4, kOpForceL1 = 5, kOpSpeed = 6, kOpDofSq = 7, kOpAbsDz = 8, kOpBelow = 9 };
constexpr int kMaxTerms = 16;
struct DevTerm {
    int op, n, src[3], idx[3];
    float w, p[8];
    int pad[3];
};
struct DevCost {
    int kind, link[4], actor[6], pad;
    float w[16];
    int n_terms, pad2[3];
    DevTerm t[kMaxTerms];
};

// Uniform (per-launch constant) structs are read through the CONSTANT address space on the GPU:
// such loads are selected as scalar loads (s_load_*, operands stay in SGPRs, zero VGPR cost)
// whatever stores the kernel performs.  launder() makes the base pointer opaque at a chosen point so
// the ~300 model constants are re-fetched from the scalar cache where they are used instead of being
// hoisted out of the time loop and parked in (then spilled from) vector registers.
#if defined(__HIP_DEVICE_COMPILE__)
#define MPPI_CONST_AS __attribute__((address_space(4)))
#else
#define MPPI_CONST_AS
#endif
typedef const MPPI_CONST_AS DevModel CModel;
typedef const MPPI_CONST_AS DevBody CBody;
typedef const MPPI_CONST_AS DevLink CLink;
typedef const MPPI_CONST_AS DevShape CShape;
typedef const MPPI_CONST_AS DevPair CPair;
typedef const MPPI_CONST_AS DevFree CFree;
typedef const MPPI_CONST_AS DevCfg CCfg;
typedef const MPPI_CONST_AS DevCost CCost;
typedef const MPPI_CONST_AS float cfloat;

// one 64-byte block of a uniform struct -> 16 SGPRs with a single scalar load
#if defined(__clang__)
typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
#else
typedef unsigned u32x16 __attribute__((vector_size(64)));
#endif
template <class B, class S>
MPPI_HD B load_block(const MPPI_CONST_AS S &src) {
    static_assert(sizeof(B) == 64 && sizeof(S) == 64, "64-byte blocks only");
    B out;
#if defined(__HIP_DEVICE_COMPILE__)
    const u32x16 v = *reinterpret_cast<const MPPI_CONST_AS u32x16 *>(&src);
    __builtin_memcpy(&out, &v, 64);
#else  // host builds (tests/hostemu): no typed access through another type - the optimiser may order such a read
       // before the float stores that fill the block (seen with g++ -O2 on a StepConsts filled just before the call)
    __builtin_memcpy(&out, &src, 64);
#endif
    return out;
}
#if defined(__HIP_DEVICE_COMPILE__)
// the same block from an LDS copy of the model: four ds_read_b128 (every lane reads the same address ->
// broadcast), in-order returns, VGPR operands (no SGPR constant-bus limit, no SGPR spills)
#define MPPI_LDS_AS __attribute__((address_space(3)))
template <class B, class S>
__device__ __forceinline__ B load_block(const MPPI_LDS_AS S &src) {
    static_assert(sizeof(B) == 64 && sizeof(S) == 64, "64-byte blocks only");
    const u32x16 v = *reinterpret_cast<const MPPI_LDS_AS u32x16 *>(&src);
    B out;
    __builtin_memcpy(&out, &v, 64);
    return out;
}
// the same block through the VECTOR memory path although the address is wave-uniform: scalar loads count on lgkmcnt together
// with LDS operations and return out of order, so a scalar load issued ahead of time is waited for by the very next LDS
// wait - it cannot be a prefetch in code that works out of LDS.  Vector loads count on vmcnt and return in order.
template <class B, class S>
__device__ __forceinline__ B load_block_vmem(const MPPI_CONST_AS S &src) {
    static_assert(sizeof(B) == 64 && sizeof(S) == 64, "64-byte blocks only");
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    unsigned zero;
    asm("v_mov_b32 %0, 0" : "=v"(zero));  // (opaque: keeps the address in a VGPR)
    const __attribute__((address_space(1))) char *p = (const __attribute__((address_space(1))) char *)(unsigned long long)(&src) + zero;
    u32x4 v[4];
    for (int j = 0; j < 4; j++) v[j] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4 *>(p + 16 * j);
    B out;
    __builtin_memcpy(&out, v, 64);
    return out;
}
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
typedef const MPPI_LDS_AS DevModel LModel;
__device__ __forceinline__ LModel *launder(LModel *p) {
    asm volatile("" : "+v"(p));
    return p;
}
#endif

template <class P>
MPPI_HD P *launder(P *p) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(p));
#endif
    return p;
}

enum { kCostNone = 0, kCostPointReach = 1, kCostPandaReach = 2, kCostBoxerPush = 3, kCostPandaPick = 4, kCostProgram = 5 };
enum { kDriveVelocity = 0, kDriveEffort = 1, kDrivePosition = 2 };

// ---- compile-time kinematic tree -------------------------------------------------------
// A parent index -1 - r names base r of the forest: r = 0 for every single-robot model and for the fixed-base forests, r > 0 for
// the further robots of an env of MOVING-base robots (mppi_hip.h, ABI 7: one floating base per tree) - so the number of bases,
// like the tree, is a compile-time fact and the instantiations of the existing trees do not change.
constexpr int topo_nbase(std::initializer_list<int> parents) {
    int n = 1;
    for (int p : parents)
        if (p < 0 && -p > n) n = -p;
    return n;
}
template <int... P>
struct Topo {
    static constexpr int NB = sizeof...(P);
    static constexpr int par[sizeof...(P) ? sizeof...(P) : 1] = {P...};
    static constexpr int NBASE = topo_nbase({P...});
};
template <int I>
struct IC {
    static constexpr int value = I;
    constexpr operator int() const { return I; }
};
template <int B, int E, class F>
MPPI_HD void static_for(F &&f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}
template <int B, int E, class F>
MPPI_HD void static_rfor(F &&f) {  // E-1 down to B
    if constexpr (B < E) {
        f(IC<E - 1>{});
        static_rfor<B, E - 1>(f);
    }
}

// ---- small vector helpers --------------------------------------------------------------
struct V3 {
    float x, y, z;
};
MPPI_HD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
MPPI_HD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
MPPI_HD V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
MPPI_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MPPI_HD V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
struct M3 {  // row-major
    float a[9];
};
MPPI_HD V3 mul(const M3 &A, V3 v) {
    return {A.a[0] * v.x + A.a[1] * v.y + A.a[2] * v.z, A.a[3] * v.x + A.a[4] * v.y + A.a[5] * v.z, A.a[6] * v.x + A.a[7] * v.y + A.a[8] * v.z};
}
MPPI_HD M3 mul(const M3 &A, const M3 &B) {
    M3 C;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) C.a[3 * i + j] = A.a[3 * i] * B.a[j] + A.a[3 * i + 1] * B.a[3 + j] + A.a[3 * i + 2] * B.a[6 + j];
    return C;
}
template <class F>
MPPI_HD M3 load3(F *p) {
    M3 A;
    for (int i = 0; i < 9; i++) A.a[i] = p[i];
    return A;
}
template <class F>
MPPI_HD V3 loadv(F *p) {
    return {p[0], p[1], p[2]};
}
// symmetric 3x3: xx xy xz yy yz zz
struct S3 {
    float xx, xy, xz, yy, yz, zz;
};
MPPI_HD V3 mul(const S3 &A, V3 v) { return {A.xx * v.x + A.xy * v.y + A.xz * v.z, A.xy * v.x + A.yy * v.y + A.yz * v.z, A.xz * v.x + A.yz * v.y + A.zz * v.z}; }

// quaternion xyzw -> rotation (root_state layout, reference isaacgym_wrapper.py:186-188)
MPPI_HD M3 quat_to_R(const float *q) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float n = x * x + y * y + z * z + w * w;
    float s = n > 0.f ? 2.f / n : 0.f;
    M3 R;
    R.a[0] = 1 - s * (y * y + z * z); R.a[1] = s * (x * y - z * w);     R.a[2] = s * (x * z + y * w);
    R.a[3] = s * (x * y + z * w);     R.a[4] = 1 - s * (x * x + z * z); R.a[5] = s * (y * z - x * w);
    R.a[6] = s * (x * z - y * w);     R.a[7] = s * (y * z + x * w);     R.a[8] = 1 - s * (x * x + y * y);
    return R;
}
// rotation -> quaternion xyzw, canonical sign w >= 0
MPPI_HD void R_to_quat(const M3 &Rm, float *q) {
    const float *R = Rm.a;
    float tr = R[0] + R[4] + R[8], x, y, z, w;
    if (tr > 0.f) {
        float s = sqrtf(tr + 1.f) * 2.f;
        w = 0.25f * s; x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        float s = sqrtf(1.f + R[0] - R[4] - R[8]) * 2.f;
        w = (R[7] - R[5]) / s; x = 0.25f * s; y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
    } else if (R[4] > R[8]) {
        float s = sqrtf(1.f + R[4] - R[0] - R[8]) * 2.f;
        w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = 0.25f * s; z = (R[5] + R[7]) / s;
    } else {
        float s = sqrtf(1.f + R[8] - R[0] - R[4]) * 2.f;
        w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = 0.25f * s;
    }
    if (w < 0.f) { x = -x; y = -y; z = -z; w = -w; }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

// ---- spatial quantities in world axes about the world origin ----------------------------
struct SV {  // motion (w, vO) or force (nO, f)
    V3 a, l;
};
MPPI_HD SV operator+(SV p, SV q) { return {p.a + q.a, p.l + q.l}; }
MPPI_HD float dot(SV p, SV q) { return dot(p.a, q.a) + dot(p.l, q.l); }
// general symmetric 6x6 [[I, H],[H^T, M]]
struct AI {
    S3 I;
    float H[9];
    S3 M;
};
MPPI_HD SV mul(const AI &A, SV v) {
    V3 n = mul(A.I, v.a), f = mul(A.M, v.l);
    n.x += A.H[0] * v.l.x + A.H[1] * v.l.y + A.H[2] * v.l.z;
    n.y += A.H[3] * v.l.x + A.H[4] * v.l.y + A.H[5] * v.l.z;
    n.z += A.H[6] * v.l.x + A.H[7] * v.l.y + A.H[8] * v.l.z;
    f.x += A.H[0] * v.a.x + A.H[3] * v.a.y + A.H[6] * v.a.z;
    f.y += A.H[1] * v.a.x + A.H[4] * v.a.y + A.H[7] * v.a.z;
    f.z += A.H[2] * v.a.x + A.H[5] * v.a.y + A.H[8] * v.a.z;
    return {n, f};
}
MPPI_HD void add_to(AI &A, const AI &B) {
    A.I.xx += B.I.xx; A.I.xy += B.I.xy; A.I.xz += B.I.xz; A.I.yy += B.I.yy; A.I.yz += B.I.yz; A.I.zz += B.I.zz;
    for (int i = 0; i < 9; i++) A.H[i] += B.H[i];
    A.M.xx += B.M.xx; A.M.xy += B.M.xy; A.M.xz += B.M.xz; A.M.yy += B.M.yy; A.M.yz += B.M.yz; A.M.zz += B.M.zz;
}
// A -= U U^T * s
MPPI_HD void rank1_sub(AI &A, SV U, float s) {
    V3 n = s * U.a, f = s * U.l;
    A.I.xx -= U.a.x * n.x; A.I.xy -= U.a.x * n.y; A.I.xz -= U.a.x * n.z; A.I.yy -= U.a.y * n.y; A.I.yz -= U.a.y * n.z; A.I.zz -= U.a.z * n.z;
    A.H[0] -= U.a.x * f.x; A.H[1] -= U.a.x * f.y; A.H[2] -= U.a.x * f.z;
    A.H[3] -= U.a.y * f.x; A.H[4] -= U.a.y * f.y; A.H[5] -= U.a.y * f.z;
    A.H[6] -= U.a.z * f.x; A.H[7] -= U.a.z * f.y; A.H[8] -= U.a.z * f.z;
    A.M.xx -= U.l.x * f.x; A.M.xy -= U.l.x * f.y; A.M.xz -= U.l.x * f.z; A.M.yy -= U.l.y * f.y; A.M.yz -= U.l.y * f.z; A.M.zz -= U.l.z * f.z;
}

// Controls of horizon step t for sample k: u = clamp(U_t + eps), effective perturbation du = u - U_t
// (stored by `leader` lanes), control-cost increment.  All loads are issued before the first use so the
// wave waits once, not once per control dimension; everything is branch-free over the padded kMaxNu.
// Hardware square root / reciprocal square root / reciprocal (1 ulp; the reciprocal gets one Newton step).  The IEEE
// library versions cost 22 (sqrtf), 32 (1/sqrtf) and ~10 (division) instructions each on gfx950 - per contact point, per
// Cholesky pivot, per cost term - for a last-bit difference that is far inside the tolerances of this path.
MPPI_HD float fsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
MPPI_HD float frsqrt(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.f / sqrtf(x);
#endif
}
MPPI_HD float frcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf(x);
    return r * (2.f - x * r);
#else
    return 1.f / x;
#endif
}

// atan2 and asin for the tilt term of the arm objectives, from the hardware reciprocal / square root and short polynomials
// (cephes atanf / asinf kernels, |error| < 2e-7 rad - tests/test_hostemu_parity.py): ~30 / ~20 instructions instead of the
// ~50 / ~35 of the library sequences.  Experiment builds only (MPPI_BUILD_VARIANT=xmppi_fast_atan): measured
// on the headline kernel, see the variant table there.
MPPI_HD float fast_atan_pos(float a) {  // a >= 0
    // cephes atanf: two range reductions to |t| <= tan(pi/8), then a degree-7 odd polynomial
    const bool big = a > 2.414213562373095f, mid = a > 0.4142135623730950f;
    const float t = big ? -frcp(a) : (mid ? (a - 1.f) * frcp(a + 1.f) : a);
    const float base = big ? 1.5707963267948966f : (mid ? 0.7853981633974483f : 0.f);
    const float z = t * t;
    const float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * t + t;
    return base + p;
}
MPPI_HD float fast_atan2(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    float r = ax > 0.f || ay > 0.f ? fast_atan_pos(ay * frcp(fmaxf(ax, 1e-30f))) : 0.f;
    r = ax * 4.e18f < ay ? 1.5707963267948966f : r;  // (x == 0 or the ratio overflows)
    r = x < 0.f ? 3.14159265358979f - r : r;
    return y < 0.f ? -r : r;
}
MPPI_HD float fast_asin(float x) {  // |x| <= 1
    const float a = fabsf(x);
    const bool big = a > 0.5f;
    const float z = big ? 0.5f * (1.f - a) : a * a;
    const float s = big ? fsqrt(z) : a;
    const float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * s + s;
    const float r = big ? 1.5707963267948966f - 2.f * p : p;
    return x < 0.f ? -r : r;
}

// clamp to [lo, hi]: one v_med3_f32 on the device (fminf(fmaxf()) costs two operations plus a canonicalisation each)
MPPI_HD float clampf(float v, float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(v, lo, hi);
#else
    return fminf(fmaxf(v, lo), hi);
#endif
}

// MAXC: compile-time bound of the control dimension (nu <= number of driven bodies of the kinematic tree), so the
// unrolled loops stop at 7 for the Panda instead of kMaxNu = 12; u[] is still written up to kMaxNu.
// The work is split in a LOAD half (nominal row, this sample's noise, prior row of step t) and an APPLY half, so that a
// rollout can request step t+1's rows before it simulates step t: the ~1 us HBM latency of the streamed noise then
// hides under the step instead of stalling the (only) wavefront of the SIMD once per horizon step.
template <int MAXC>
struct ControlRows {
    float Ut[MAXC], e[MAXC], pr[MAXC];
};
template <int MAXC = kMaxNu>
MPPI_HD void load_controls(CCfg &cfg, const float *U, const float *eps, const float *prior, int t, int k, ControlRows<MAXC> &r) {
    const int K = cfg.K, nu = cfg.nu;
    // 32-bit element indices (H * nu * K < 2^31 is checked by pack_config): no 64-bit scalar multiplies per load.
    // The prior row is loaded by every lane when there is one (uniform address, uniform condition) and selected per
    // lane in apply_controls - a per-lane `is_prior ? prior[..] : 0` would put each load under its own exec-mask branch.
    const bool has_prior = prior != nullptr;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        const int cc = c < nu ? c : nu - 1;  // keep the address valid; the lane is masked in apply_controls
        const unsigned row = (unsigned)(t * nu + cc);
        r.Ut[c] = U[row];
        r.e[c] = eps[row * (unsigned)K + (unsigned)k];
        r.pr[c] = 0.f;
    }
    if (has_prior) {  // ONE uniform branch for the whole row (rare: use_priors)
#pragma unroll
        for (int c = 0; c < MAXC; c++) r.pr[c] = prior[t * nu + (c < nu ? c : nu - 1)];
    }
}
template <int MAXC = kMaxNu>
MPPI_HD float apply_controls(CCfg &cfg, const ControlRows<MAXC> &r, int t, int k, bool is_null, bool is_prior, bool /*leader*/, float *du, float *u) {
    const int K = cfg.K, nu = cfg.nu;
#pragma unroll
    for (int c = MAXC; c < kMaxNu; c++) u[c] = 0.f;
    const CtrlBlock lo = load_block<CtrlBlock>(cfg.u_min), hi = load_block<CtrlBlock>(cfg.u_max), is = load_block<CtrlBlock>(cfg.inv_sigma);
    float ctrl = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        float v = r.Ut[c] + r.e[c];
        if (is_null) v = 0.f;
        if (is_prior) v = r.pr[c];
        v = clampf(v, lo.v[c], hi.v[c]);
        const bool on = c < nu;
        u[c] = on ? v : 0.f;
        const float d = v - r.Ut[c];
        // (`leader` is not consulted: lanes that share a sample hold the same du and store it to the same address - plain
        // stores under a uniform condition instead of one exec-mask region per control)
        if (on) du[(unsigned)(t * nu + c) * (unsigned)K + (unsigned)k] = d;
        const float term = r.Ut[c] * d * is.v[c];  // inv_sigma is zero beyond nu
        ctrl += cfg.lambda * (cfg.noise_abs_cost ? fabsf(term) : term);
    }
    return ctrl;
}
template <int MAXC = kMaxNu>
MPPI_HD float sample_controls(CCfg &cfg, const float *U, const float *eps, const float *prior, int t, int k, bool is_null, bool is_prior,
                              bool leader, float *du, float *u) {
    ControlRows<MAXC> r;
    load_controls<MAXC>(cfg, U, eps, prior, t, k, r);
    return apply_controls<MAXC>(cfg, r, t, k, is_null, is_prior, leader, du, u);
}

// World pose of every moving body for joint positions q (z-framed joints).
template <class T>
struct Pose {
    M3 R[T::NB ? T::NB : 1];
    V3 p[T::NB ? T::NB : 1];
    int jt[T::NB ? T::NB : 1];  // joint types (wave-uniform), cached with the pose
    M3 Rb;
    V3 pb;
    M3 Rx[T::NBASE > 1 ? T::NBASE - 1 : 1];  // bases 1.. of a forest of moving-base robots (base 0: Rb, pb)
    V3 px[T::NBASE > 1 ? T::NBASE - 1 : 1];
    template <int r> MPPI_HD const M3 &base_R() const { if constexpr (r == 0) return Rb; else return Rx[r - 1]; }
    template <int r> MPPI_HD const V3 &base_p() const { if constexpr (r == 0) return pb; else return px[r - 1]; }
    template <int r> MPPI_HD M3 &base_R() { if constexpr (r == 0) return Rb; else return Rx[r - 1]; }
    template <int r> MPPI_HD V3 &base_p() { if constexpr (r == 0) return pb; else return px[r - 1]; }
};
// base r of the model: root row, composite inertia of the root-link cluster
template <int r, class M> MPPI_HD int base_actor(M &m) { if constexpr (r == 0) return m.robot_actor; else return m.xbase_actor[r - 1]; }
template <int r, class M> MPPI_HD float base_mass(M &m) { if constexpr (r == 0) return m.base_m; else return m.xbase_m[r - 1]; }
constexpr int base_of_parent(int par) { return par < 0 ? -1 - par : 0; }

#define MPPI_LAMBDA __attribute__((always_inline))

// sin and cos with a 3-term Cody-Waite reduction by pi/2 and cephes minimax polynomials on
// [-pi/4, pi/4] (~1 ulp for |x| < 1e4).  Replaces ocml's sincosf, whose large-argument path keeps a
// private array in scratch memory; identical arithmetic on host and device.
MPPI_HD void fast_sincos(float x, float &s, float &c) {
    float k = rintf(x * 0.636619772367581343f);
    float r = fmaf(k, -1.5703125f, x);
    r = fmaf(k, -4.837512969970703125e-4f, r);
    r = fmaf(k, -7.54978995489188216e-8f, r);
    int ki = (int)k;
    float r2 = r * r;
    float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f), r2 * r, r);
    float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f), r2 * r2, fmaf(-0.5f, r2, 1.0f));
    float ss = (ki & 1) ? cp : sp;
    float cc = (ki & 1) ? sp : cp;
    s = (ki & 2) ? -ss : ss;
    c = ((ki + 1) & 2) ? -cc : cc;
}

template <class T, class M>
MPPI_HD void forward_kinematics_base(M &m, const float *q, Pose<T> &P);

template <class T, class M>
MPPI_HD void forward_kinematics(M &m, const float *root, const float *q, Pose<T> &P) {
    static_for<0, T::NBASE>([&](auto rc) MPPI_LAMBDA {
        constexpr int r = rc;
        const float *rs = root + 13 * base_actor<r>(m);
        P.template base_p<r>() = loadv(rs);
        P.template base_R<r>() = quat_to_R(rs + 3);
    });
    forward_kinematics_base<T>(m, q, P);
}

// body poses from the base pose already stored in P.Rb / P.pb
template <class T, class M>
MPPI_HD void forward_kinematics_base(M &m, const float *q, Pose<T> &P) {
    static_for<0, T::NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const BodyK0 b = load_block<BodyK0>(m.b[i].k0);
        P.jt[i] = b.jtype;
        const M3 &Rp = par < 0 ? P.template base_R<base_of_parent(par)>() : P.R[par < 0 ? 0 : par];
        const V3 pp = par < 0 ? P.template base_p<base_of_parent(par)>() : P.p[par < 0 ? 0 : par];
        M3 Rt0;
        for (int j = 0; j < 9; j++) Rt0.a[j] = b.rt(j);
        M3 RT = mul(Rp, Rt0);
        V3 pw = pp + mul(Rp, V3{b.pt(0), b.pt(1), b.pt(2)});
        if (b.jtype == 0) {  // revolute about local z: R = RT * Rz(q)
            float s, c;
            fast_sincos(q[i], s, c);
            M3 R;
            for (int r = 0; r < 3; r++) {
                R.a[3 * r + 0] = c * RT.a[3 * r] + s * RT.a[3 * r + 1];
                R.a[3 * r + 1] = c * RT.a[3 * r + 1] - s * RT.a[3 * r];
                R.a[3 * r + 2] = RT.a[3 * r + 2];
            }
            P.R[i] = R;
            P.p[i] = pw;
        } else {  // prismatic along local z
            V3 az = {RT.a[2], RT.a[5], RT.a[8]};
            P.R[i] = RT;
            P.p[i] = pw + q[i] * az;
        }
    });
}

// joint motion subspace (world axes, about the world origin) of body i
template <class T, int i, class M>
MPPI_HD SV joint_subspace(M &m, const Pose<T> &P) {
    V3 az = {P.R[i].a[2], P.R[i].a[5], P.R[i].a[8]};
    if (P.jt[i] == 0) return {az, cross(P.p[i], az)};
    return {{0.f, 0.f, 0.f}, az};
}

// One articulated-body solve (Featherstone ABA, world-frame form) with the implicit joint drive
// folded into the joint-space inertia: d_i = S_i^T IA_i S_i + kdh[i];  tau_exp = explicit part.
// Register diet: only poses (12) and velocities (6) per body are carried between the passes; the
// rigid inertia, bias force, S and c of a body are formed when the backward sweep reaches it.
template <class T>
MPPI_HD void aba_world(CModel &m, const Pose<T> &P, const float *qd, const float *tau_exp, const float *kdh, float *qdd) {
    constexpr int NB = T::NB;
    SV v[NB], U[NB];
    AI acc[NB];   // children's articulated inertia, accumulated on the parent (live only while pending)
    SV pacc[NB];
    float invd[NB], u[NB];
    bool has_acc[NB];
    // pass 1: spatial velocities, root to leaves
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        SV S = joint_subspace<T, i>(m, P);
        SV sj = {qd[i] * S.a, qd[i] * S.l};
        if constexpr (par < 0) v[i] = sj;
        else v[i] = v[par < 0 ? 0 : par] + sj;
        has_acc[i] = false;
    });
    // pass 2: articulated inertias, leaves to root (world frame: propagation is an addition)
    static_rfor<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        const BodyK1 b = load_block<BodyK1>(m.b[i].k1);
        const M3 &R = P.R[i];
        SV S = joint_subspace<T, i>(m, P);
        // rigid inertia about the world origin: I_O = R Ic R^T + m(|cw|^2 1 - cw cw^T), h = m cw
        V3 h = mul(R, V3{b.hb(0), b.hb(1), b.hb(2)}) + b.m * P.p[i];
        float T9[9];  // T = R * Ic
        for (int r = 0; r < 3; r++) {
            float r0 = R.a[3 * r], r1 = R.a[3 * r + 1], r2 = R.a[3 * r + 2];
            T9[3 * r + 0] = r0 * b.Ic(0) + r1 * b.Ic(1) + r2 * b.Ic(2);
            T9[3 * r + 1] = r0 * b.Ic(1) + r1 * b.Ic(3) + r2 * b.Ic(4);
            T9[3 * r + 2] = r0 * b.Ic(2) + r1 * b.Ic(4) + r2 * b.Ic(5);
        }
        float invm = b.invm;
        V3 cw = invm * h;
        float hh = dot(h, cw);
        AI A;
        A.I.xx = T9[0] * R.a[0] + T9[1] * R.a[1] + T9[2] * R.a[2] + hh - h.x * cw.x;
        A.I.xy = T9[0] * R.a[3] + T9[1] * R.a[4] + T9[2] * R.a[5] - h.x * cw.y;
        A.I.xz = T9[0] * R.a[6] + T9[1] * R.a[7] + T9[2] * R.a[8] - h.x * cw.z;
        A.I.yy = T9[3] * R.a[3] + T9[4] * R.a[4] + T9[5] * R.a[5] + hh - h.y * cw.y;
        A.I.yz = T9[3] * R.a[6] + T9[4] * R.a[7] + T9[5] * R.a[8] - h.y * cw.z;
        A.I.zz = T9[6] * R.a[6] + T9[7] * R.a[7] + T9[8] * R.a[8] + hh - h.z * cw.z;
        A.H[0] = 0.f;  A.H[1] = -h.z; A.H[2] = h.y;
        A.H[3] = h.z;  A.H[4] = 0.f;  A.H[5] = -h.x;
        A.H[6] = -h.y; A.H[7] = h.x;  A.H[8] = 0.f;
        A.M = {b.m, 0.f, 0.f, b.m, 0.f, b.m};
        // bias force v x* (I v) of the rigid body
        V3 n = mul(A.I, v[i].a) + cross(h, v[i].l);
        V3 f = b.m * v[i].l + cross(v[i].a, h);
        SV pA = {cross(v[i].a, n) + cross(v[i].l, f), cross(v[i].a, f)};
        if (has_acc[i]) {
            add_to(A, acc[i]);
            pA = pA + pacc[i];
        }
        U[i] = mul(A, S);
        float d = dot(S, U[i]) + kdh[i];
        invd[i] = frcp(d);
        u[i] = tau_exp[i] - dot(S, pA);
        if constexpr (par >= 0) {
            // c = v_parent x (S qd);  pa = pA + IA c + U (u - U.c)/d;  Ia = IA - U U^T / d
            const SV vp = v[par < 0 ? 0 : par];
            SV sj = {qd[i] * S.a, qd[i] * S.l};
            SV c = {cross(vp.a, sj.a), cross(vp.a, sj.l) + cross(vp.l, sj.a)};
            SV Ic_ = mul(A, c);
            float k = (u[i] - dot(U[i], c)) * invd[i];
            SV pa = {pA.a + Ic_.a + k * U[i].a, pA.l + Ic_.l + k * U[i].l};
            rank1_sub(A, U[i], invd[i]);
            constexpr int pj = par < 0 ? 0 : par;
            if (has_acc[pj]) {
                add_to(acc[pj], A);
                pacc[pj] = pacc[pj] + pa;
            } else {
                acc[pj] = A;
                pacc[pj] = pa;
                has_acc[pj] = true;
            }
        }
    });
    // pass 3: accelerations, root to leaves.  Gravity = fictitious base acceleration -g.
    SV a[NB];
    SV a0 = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (m.gravity_on) a0.l = {-m.g[0], -m.g[1], -m.g[2]};
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        SV S = joint_subspace<T, i>(m, P);
        SV ap = a0;
        if constexpr (par >= 0) {
            const SV vp = v[par < 0 ? 0 : par];
            SV sj = {qd[i] * S.a, qd[i] * S.l};
            SV c = {cross(vp.a, sj.a), cross(vp.a, sj.l) + cross(vp.l, sj.a)};
            ap = a[par < 0 ? 0 : par] + c;
        }
        float dd = (u[i] - dot(U[i], ap)) * invd[i];
        qdd[i] = dd;
        a[i] = {ap.a + dd * S.a, ap.l + dd * S.l};
    });
}

// apply_robot_cmd: control u[nu] -> per-DOF drive target (reference isaacgym_wrapper.py:524-572)
template <class T, class M>
MPPI_HD void cmd_map(M &m, const float *u, float *target) {
    static_for<0, T::NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        const CmdBlock b = load_block<CmdBlock>(m.b[i].cmd);
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < (T::NB < kMaxNu ? T::NB : kMaxNu); c++) t += b.v[c] * u[c];  // nu <= NB (mppi_pack.hpp); rows and u are zero beyond nu
        target[i] = t;
    });
}

// Inelastic joint limit: the position is clamped and the velocity becomes the displacement that actually happened over the
// step, (x_new - x_old) / h - never pointing back out of the range.  (Zeroing the velocity at the stop is a jump: a joint that
// reaches its limit one substep earlier or later - 1e-9 rad decide - differs by its full speed for that substep; in the
// gripper scene that was the largest single source of fp32 / fp64 disagreement.)
MPPI_HD void joint_limit(float x_old, float &x, float &v, float lower, float upper, float inv_h) {
    if (x < lower) { x = lower; v = fminf((lower - x_old) * inv_h, 0.f); }
    if (x > upper) { x = upper; v = fmaxf((upper - x_old) * inv_h, 0.f); }
}

// One simulator step dt = substeps * h (semi-implicit Euler, implicit velocity-level drive,
// drive-force clamp by one re-solve, velocity clamp, inelastic joint limits).  SURVEY.md B.
template <class T>
MPPI_HD void step(CModel &m0, const float *root, float *q, float *qd, const float *target) {
    constexpr int NB = T::NB;
    CModel *mp = &m0;
    // position mode (reference isaacgym_wrapper.py:571-572): apply_robot_cmd overwrites the DOF state with the command
    if (m0.drive_mode == kDrivePosition)
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA { q[ic] = target[ic]; qd[ic] = 0.f; });
    for (int s = 0; s < m0.substeps; s++) {
        CModel &m = *launder(mp);  // re-fetch model constants per substep (see launder())
        // drives, implicit in the velocity: velocity kd (target - qd), effort target - kd qd, position kp (target - q) - kd qd with
        // q taken at the END of the substep, which is the same form with the damping kd + h kp (oracle: drive_damping)
        const bool pos = m.drive_mode == kDrivePosition;
        const float h = m.h, kp = pos ? m.kp : 0.f, kd = m.kd + h * kp;
        Pose<T> P;
        forward_kinematics<T>(m, root, q, P);
        float tau[NB], kdh[NB], qdd[NB], ff[NB], vs[NB];
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            ff[i] = m.drive_mode == kDriveEffort ? target[i] : (pos ? kp * (target[i] - q[i]) : 0.f);
            vs[i] = m.drive_mode == kDriveVelocity ? target[i] : 0.f;
            tau[i] = ff[i] + kd * (vs[i] - qd[i]);
            kdh[i] = kd * h;
        });
        aba_world<T>(m, P, qd, tau, kdh, qdd);
        bool any = false;
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            float lim = m.b[i].k1.effort;
            float tt = ff[i] + kd * (vs[i] - qd[i] - h * qdd[i]);
            if (lim > 0.f && fabsf(tt) > lim) {
                any = true;
                tau[i] = tt > 0.f ? lim : -lim;
                kdh[i] = 0.f;
            }
        });
        if (any) aba_world<T>(*launder(mp), P, qd, tau, kdh, qdd);
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            const BodyK1 b = load_block<BodyK1>(m.b[i].k1);
            float v = qd[i] + h * qdd[i];
            if (b.vmax > 0.f) v = fminf(fmaxf(v, -b.vmax), b.vmax);
            float x = q[i] + h * v;
            const float lo = m.b[i].k0.lower, hi = m.b[i].k0.upper;
            if (lo > -INFINITY || hi < INFINITY) joint_limit(q[i], x, v, lo, hi, 1.f / h);
            q[i] = x;
            qd[i] = v;
        });
    }
}

// world pose of link l (R, p) from body poses.  The body a link is welded to is a run-time
// (wave-uniform) index; it is resolved by a 0/1-weighted blend over the compile-time body list
// instead of a select chain, because the optimiser turns "select between array elements" into an
// indexed load and then keeps the whole Pose in scratch memory.
template <class T, class M>
MPPI_HD void link_pose(M &m, const Pose<T> &P, int l, M3 &R, V3 &p) {
    auto &L = m.l[l];
    const int body = L.body;
    float wb = (T::NBASE > 1 ? body == -1 : body < 0) ? 1.f : 0.f;
    M3 Rb;
    for (int j = 0; j < 9; j++) Rb.a[j] = wb * P.Rb.a[j];
    V3 pb = wb * P.pb;
    static_for<0, T::NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        const float w = body == i ? 1.f : 0.f;
        for (int j = 0; j < 9; j++) Rb.a[j] += w * P.R[i].a[j];
        pb = pb + w * P.p[i];
    });
    if constexpr (T::NBASE > 1)   // links welded to a further base of the forest (body = -1 - r)
        static_for<1, T::NBASE>([&](auto rc) MPPI_LAMBDA {
            constexpr int r = rc;
            const float w = body == -1 - r ? 1.f : 0.f;
            for (int j = 0; j < 9; j++) Rb.a[j] += w * P.Rx[r - 1].a[j];
            pb = pb + w * P.px[r - 1];
        });
    R = mul(Rb, load3(L.R));
    p = pb + mul(Rb, loadv(L.p));
}

MPPI_HD float clamp1(float x) { return fminf(fmaxf(x, -1.f), 1.f); }

// ---- cost programs (MPPI_COST_PROGRAM) ------------------------------------------------------------------
// cost = sum_i w_i * measurement_i over the term list of DevCost (every Objective of the reference's examples is such a
// sum; include/mppi_hip.h MPPI_OP_*).  The term list is uniform over the wavefront, so every branch below is a scalar
// branch.  `E` answers what the env looks like for THIS sample: actor rows (position / quaternion / velocity of the robot
// base, the free actors, the static actors) and net contact forces; link poses come from the sample's kinematics `P`.
// `link_of(l, R, p)` yields the world pose of robot link l from whatever kinematics the caller holds.
template <class T, class LP, class E>
MPPI_HD float program_cost_with(CCost &c, const float *q, const float *qd, const LP &link_of, const E &env) {
    float total = 0.f;
    for (int it = 0; it < c.n_terms; it++) {
        auto &t = c.t[it];
        auto point = [&](int a) MPPI_LAMBDA {
            const int src = t.src[a], idx = t.idx[a];
            if (src == kSrcLink) {
                M3 R;
                V3 p;
                link_of(idx, R, p);
                return p;
            }
            if (src == kSrcActor) return env.vec(idx, 0);
            if (src == kSrcDofXY) return V3{q[0], q[T::NB > 1 ? 1 : 0], 0.f};
            if (src == kSrcConst) return env.constant_point(t.p[0], t.p[1], t.p[2]);
            return V3{0.f, 0.f, 0.f};
        };
        float v = 0.f;
        const int op = t.op;
        if (op == kOpDist) {
            const V3 d = point(0) - point(1);
            v = fsqrt(d.x * d.x + d.y * d.y + (t.n > 2 ? d.z * d.z : 0.f));
        } else if (op == kOpTilt) {
            M3 R;
            V3 p;
            link_of(t.idx[0], R, p);
            const float a0 = atan2f(R.a[7], -R.a[8]), a1 = asinf(clamp1(R.a[6]));  // see PANDA_REACH
            v = fsqrt(a0 * a0 + a1 * a1);
        } else if (op == kOpYawAbs) {
            float qq[4];
            env.quat(t.idx[0], qq);
            v = fabsf(atan2f(2.f * (qq[3] * qq[2] + qq[0] * qq[1]), qq[3] * qq[3] + qq[0] * qq[0] - qq[1] * qq[1] - qq[2] * qq[2]) - t.p[3]);
        } else if (op == kOpAlign) {
            const V3 b = point(1), a = point(0) - b, cc = point(2) - b;
            v = (a.x * cc.x + a.y * cc.y) * frcp(fsqrt(a.x * a.x + a.y * a.y) * fsqrt(cc.x * cc.x + cc.y * cc.y)) + 1.f;
        } else if (op == kOpForceL1) {
            v = fabsf(env.cf(t.idx[0], 0)) + (t.n > 1 ? fabsf(env.cf(t.idx[0], 1)) : 0.f) + (t.n > 2 ? fabsf(env.cf(t.idx[0], 2)) : 0.f);
        } else if (op == kOpSpeed) {
            const V3 u = env.vec(t.idx[0], 7);
            v = fsqrt(u.x * u.x + (t.n > 1 ? u.y * u.y : 0.f) + (t.n > 2 ? u.z * u.z : 0.f));
        } else if (op == kOpDofSq) {
            const int lo = t.idx[0], hi = t.idx[1], nref = t.idx[2];
            static_for<0, T::NB>([&](auto ic) MPPI_LAMBDA {
                constexpr int i = ic;
                if (i >= lo && i < hi) {
                    const int j = i - lo;
                    const float x = (t.n == 0 ? q[i] : qd[i]) - (j < nref ? t.p[j < 8 ? j : 7] : 0.f);
                    v += x * x;
                }
            });
        } else if (op == kOpAbsDz) {
            v = fabsf(point(0).z - point(1).z);
        } else if (op == kOpBelow) {
            v = fmaxf(t.p[3] - point(0).z, 0.f);
        }
        total += t.w * v;
    }
    return total;
}
template <class T, class M, class E>
MPPI_HD float program_cost(M &m, CCost &c, const float *q, const float *qd, const Pose<T> &P, const E &env) {
    return program_cost_with<T>(c, q, qd, [&](int l, M3 &R, V3 &p) MPPI_LAMBDA { link_pose<T>(m, P, l, R, p); }, env);
}
// env of a fixed-base contact-free scene: every actor row is the static x0 row, nothing touches anything
struct StaticEnv {
    const float *root;
    MPPI_HD V3 vec(int actor, int off) const { return loadv(root + 13 * actor + off); }
    MPPI_HD void quat(int actor, float *qq) const {
        for (int j = 0; j < 4; j++) qq[j] = root[13 * actor + 3 + j];
    }
    MPPI_HD float cf(int, int) const { return 0.f; }
    MPPI_HD V3 constant_point(float x, float y, float z) const { return V3{x, y, z}; }
};

// Fused stage cost (DevCost.kind) for a given pose.  See include/mppi_hip.h for the reference Objective each restates.
template <class T, class M>
MPPI_HD float stage_cost_pose(M &m, CCost &c, const float *root, const float *q, const Pose<T> &P) {
    if (c.kind == kCostPointReach) {
        float gx = c.actor[0] >= 0 ? root[13 * c.actor[0]] : c.w[1];
        float gy = c.actor[0] >= 0 ? root[13 * c.actor[0] + 1] : c.w[2];
        float dx = q[0] - gx, dy = q[T::NB > 1 ? 1 : 0] - gy;
        return c.w[0] * sqrtf(dx * dx + dy * dy);
    }
    if (c.kind == kCostPandaReach) {
        M3 R;
        V3 p;
        link_pose<T>(m, P, c.link[0], R, p);
        V3 d = p - loadv(root + 13 * c.actor[0]);
        float dist = sqrtf(dot(d, d));
        // The reference feeds the xyzw link quaternion to pytorch3d's (r,i,j,k) API and takes the first
        // two ZYX Euler angles (examples/panda/planner.py:30-32).  For a unit quaternion the permuted
        // matrix entries are linear in the true rotation: M00 = -R22, M10 = R21, M20 = -R20.
        float a0 = atan2f(R.a[7], -R.a[8]);
        float a1 = asinf(clamp1(R.a[6]));
        return c.w[0] * dist + c.w[1] * sqrtf(a0 * a0 + a1 * a1);
    }
    return 0.f;
}

template <class T>
MPPI_HD float stage_cost(CModel &m, CCost &c, const float *root, const float *q, const float *qd) {
    if (c.kind == kCostProgram) {
        Pose<T> P;
        forward_kinematics<T>(m, root, q, P);
        return program_cost<T>(m, c, q, qd, P, StaticEnv{root});
    }
    if (c.kind != kCostPandaReach) {
        Pose<T> none;  // not read by the pose-free costs
        return stage_cost_pose<T>(m, c, root, q, none);
    }
    Pose<T> P;
    forward_kinematics<T>(m, root, q, P);
    return stage_cost_pose<T>(m, c, root, q, P);
}

// Whole-horizon rollout of ONE sample (lane).  eps/du are sample-minor: [(t*nu+c)*K + k].
// Returns S_k = sum_t gamma^t c_t + lambda * sum_t U_t^T Sigma^-1 du_t   (SURVEY.md A).
template <class T>
MPPI_HD float rollout_sample(CModel &m0, CCfg &cfg0, CCost &cost0, const float *dof0, const float *root, const float *U, const float *eps,
                             const float *prior, float *du, float *viz, int k) {
    constexpr int NB = T::NB;
    const int K = cfg0.K, nu = cfg0.nu, H = cfg0.H;
    const int g = cfg0.k_offset + k;
    const bool is_null = cfg0.sample_null_action && g == cfg0.k_total - 1;
    const bool is_prior = cfg0.use_priors && prior != nullptr && g == cfg0.k_total - 2;
    float q[NB], qd[NB], target[NB], u[kMaxNu];
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        q[i] = dof0[2 * i];
        qd[i] = dof0[2 * i + 1];
    });
    float S = 0.f, ctrl = 0.f, disc = 1.f;
    CModel *mp = &m0;
    CCfg *cp = &cfg0;
    CCost *kp = &cost0;
    for (int t = 0; t < H; t++) {
        CCfg &cfg = *launder(cp);
        ctrl += sample_controls<(T::NB < kMaxNu ? T::NB : kMaxNu)>(cfg, U, eps, prior, t, k, is_null, is_prior, true, du, u);
        cmd_map<T>(*launder(mp), u, target);
        step<T>(*mp, root, q, qd, target);
        S += disc * stage_cost<T>(*launder(mp), *launder(kp), root, q, qd);
        disc *= cfg.gamma;
        if (cfg.want_rollouts && viz != nullptr) {
            CModel &m = *launder(mp);
            Pose<T> P;
            forward_kinematics<T>(m, root, q, P);
            M3 R;
            V3 p;
            link_pose<T>(m, P, cfg.viz_link, R, p);
            // sample-minor [H][3][K]: three full-line coalesced stores per wave
            viz[((size_t)t * 3 + 0) * K + k] = p.x;
            viz[((size_t)t * 3 + 1) * K + k] = p.y;
            viz[((size_t)t * 3 + 2) * K + k] = p.z;
        }
    }
    return S + ctrl;
}

// rigid_body_state rows [n_rb][13] + net_contact_force [n_rb][3] of one env, reference layout.
template <class T>
MPPI_HD void rigid_body_state(CModel &m, const float *root, const float *q, const float *qd, float *rb, float *cf) {
    constexpr int NB = T::NB;
    Pose<T> P;
    forward_kinematics<T>(m, root, q, P);
    // body spatial velocities (world frame, about the world origin)
    SV v[NB ? NB : 1];
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        SV S = joint_subspace<T, i>(m, P);
        SV sj = {qd[i] * S.a, qd[i] * S.l};
        if constexpr (par < 0) v[i] = sj;
        else v[i] = v[par < 0 ? 0 : par] + sj;
    });
    for (int a = 0; a < m.n_actors; a++) {
        if (a == m.robot_actor) continue;
        float *o = rb + 13 * m.actor_first_rb[a];
        for (int j = 0; j < 13; j++) o[j] = root[13 * a + j];
    }
    for (int l = 0; l < m.nl; l++) {
        M3 R;
        V3 p;
        link_pose<T>(m, P, l, R, p);
        SV vb = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
            constexpr int i = ic;
            const float w = m.l[l].body == i ? 1.f : 0.f;
            vb = {vb.a + w * v[i].a, vb.l + w * v[i].l};
        });
        V3 lv = vb.l + cross(vb.a, p);  // velocity of the link origin
        float *o = rb + 13 * (m.robot_first_rb + l);
        o[0] = p.x; o[1] = p.y; o[2] = p.z;
        R_to_quat(R, o + 3);
        o[7] = lv.x; o[8] = lv.y; o[9] = lv.z;
        o[10] = vb.a.x; o[11] = vb.a.y; o[12] = vb.a.z;
    }
    if (cf != nullptr)
        for (int j = 0; j < 3 * m.n_rb; j++) cf[j] = 0.f;
}

// the rigid-body row (position, quaternion xyzw, linear and angular velocity: 13 floats) of ONE robot link of one env
template <class T>
MPPI_HD void rigid_body_link(CModel &m, const float *root, const float *q, const float *qd, int l, float *o) {
    constexpr int NB = T::NB;
    Pose<T> P;
    forward_kinematics<T>(m, root, q, P);
    SV v[NB ? NB : 1];
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        constexpr int par = T::par[i];
        SV S = joint_subspace<T, i>(m, P);
        SV sj = {qd[i] * S.a, qd[i] * S.l};
        if constexpr (par < 0) v[i] = sj;
        else v[i] = v[par < 0 ? 0 : par] + sj;
    });
    M3 R;
    V3 p;
    link_pose<T>(m, P, l, R, p);
    SV vb = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    static_for<0, NB>([&](auto ic) MPPI_LAMBDA {
        constexpr int i = ic;
        const float w = m.l[l].body == i ? 1.f : 0.f;
        vb = {vb.a + w * v[i].a, vb.l + w * v[i].l};
    });
    V3 lv = vb.l + cross(vb.a, p);
    o[0] = p.x; o[1] = p.y; o[2] = p.z;
    R_to_quat(R, o + 3);
    o[7] = lv.x; o[8] = lv.y; o[9] = lv.z;
    o[10] = vb.a.x; o[11] = vb.a.y; o[12] = vb.a.z;
}

}  // namespace mppi
