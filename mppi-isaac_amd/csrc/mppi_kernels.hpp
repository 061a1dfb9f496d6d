// mppi_kernels.hpp - every gfx950 kernel of the backend, the context they run on and the per-topology
// launch table.  Included by the main translation unit (mppi_hip.hip: C-ABI) and by one generated
// translation unit per kinematic tree (topo_<i>.hip, written by __graft_entry__.build()), so the
// compile-time-tree instantiations build in parallel.
//
// Launch structure of one control iteration (replaces the reference's Python horizon loop with
// ~H x (gym set + simulate + fetch + 4 refresh + 15-40 torch kernels), reference
// mppiisaac/planner/mppi_isaac.py:57-69 / SURVEY.md 3.1):
//   k_rollout_quad<Topo>  one sample per 4-lane quad, persistent over the whole horizon: perturb/clamp the
//                         nominal controls, H x substeps articulated-body steps, fused stage cost, discounted
//                         sum, and - in its tail - the per-wave softmax record (beta, eta, sum w du).
//                         k_rollout<Topo> (one lane per sample) and k_rollout_scene<Topo> (contact scenes,
//                         LDS-staged frames) are the other two rollout kernels.
//   k_combine[_world]     rescales and sums per-wave / per-GPU records (the same formula joins waves, and
//                         GPUs after the RCCL all-gather), updates and shifts the nominal U, emits the
//                         action; the _world variant also steps the K = 1 world and feeds its state back.
// Buffers are sample-minor so every per-sample access of a wave is one coalesced request.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mppi_pack.hpp"
#include "mppi_scene.hpp"
#include "mppi_quad.hpp"
#include "mppi_oct.hpp"
#include "mppi_scene_quad.hpp"
#include "mppi_scene_oct.hpp"

using namespace mppi;

struct mppi_ctx;

namespace {

constexpr int kWave = 64;

// ------------------------------------------------------------------------------ wave helpers
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, kWave));
    return v;
}

// ------------------------------------------------------------------------------ kernels
template <class T>
__global__ __launch_bounds__(kWave) void k_rollout(const DevModel *__restrict__ m, const DevCfg *__restrict__ cfg,
                                                   const DevCost *__restrict__ cost, const float *__restrict__ x0_dof,
                                                   const float *__restrict__ x0_root, const float *__restrict__ U,
                                                   const float *__restrict__ eps, const float *__restrict__ prior,
                                                   float *__restrict__ du, float *__restrict__ S, float *__restrict__ viz,
                                                   float *__restrict__ partials);

// Partial record [beta, eta, N[HN]] of the 64 samples of one wave (all 64 lanes must be active):
// beta = min S, w = exp(-(S-beta)/lambda), eta = sum w, N[j] = sum_k w_k du[j][k].  The du rows are
// read back 4 at a time so the shuffle reductions of one row overlap the loads of the next.
__device__ __forceinline__ void wave_record(CCfg &cfg, float s, bool live, const float *__restrict__ du, int k, float *__restrict__ rec) {
    const int K = cfg.K, HN = cfg.H * cfg.nu;
    const bool fin = live && isfinite(s);  // NaN / Inf trajectory cost -> weight 0
    const float beta = wave_min(fin ? s : INFINITY);
    const float w = fin ? __expf(-(s - beta) * cfg.inv_lambda) : 0.f;
    const float eta = wave_sum(w);
    const int lane = threadIdx.x & (kWave - 1);
    if (lane == 0) {
        rec[0] = beta;
        rec[1] = eta;
    }
    const size_t kk = live ? (size_t)k : 0;
    int j = 0;
    for (; j + 4 <= HN; j += 4) {
        float x0 = w * du[(size_t)(j + 0) * K + kk], x1 = w * du[(size_t)(j + 1) * K + kk];
        float x2 = w * du[(size_t)(j + 2) * K + kk], x3 = w * du[(size_t)(j + 3) * K + kk];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            x0 += __shfl_xor(x0, o, kWave);
            x1 += __shfl_xor(x1, o, kWave);
            x2 += __shfl_xor(x2, o, kWave);
            x3 += __shfl_xor(x3, o, kWave);
        }
        if (lane == 0) {
            rec[2 + j] = x0; rec[3 + j] = x1; rec[4 + j] = x2; rec[5 + j] = x3;
        }
    }
    for (; j < HN; j++) {
        float x = wave_sum(w * du[(size_t)j * K + kk]);
        if (lane == 0) rec[2 + j] = x;
    }
}

// The same record for a wavefront of the kernels whose lanes share samples: SPW samples (chunk * SPW ...), sample s held
// by lanes s*LPS .. s*LPS + LPS-1 (LPS = 64 / SPW lanes per sample: 4 - quad kernels - or 8).  Instead of a 64-lane
// butterfly per row (6 shuffle steps x H*nu rows), lane l sums rows l, l + 64, ... over the SPW samples itself: the
// weights are broadcast through LDS, the du values of a row are one contiguous 64- or 32-byte read.
// `owner` = false: a helper wavefront of the workgroup (k_rollout_scene_quad with NW = 2) - it owns no samples and only keeps
// the workgroup barrier company (every wavefront of a workgroup must reach every __syncthreads()).
// `slot` >= 0: the sample slot (0 .. SPW-1) this lane belongs to, for kernels whose samples are not laid out as consecutive lane
// groups (the octet layout of the contact-free rollout, mppi_oct.hpp: two samples interleaved in every 16-lane row).
template <int SPW = 16>
__device__ __forceinline__ void quad_record(CCfg &cfg, float s, bool live_leader, const float *__restrict__ du, int k0, float *__restrict__ rec,
                                            bool owner = true, int slot = -1) {
    constexpr int LPS = kWave / SPW;
    __shared__ float s_w[SPW];
    if (!owner) {
        MPPI_BARRIER(5);
        return;
    }
    const int K = cfg.K, HN = cfg.H * cfg.nu;
    const int lane = threadIdx.x & (kWave - 1);
    const bool fin = live_leader && isfinite(s);
    const float beta = wave_min(fin ? s : INFINITY);
    const float w = fin ? __expf(-(s - beta) * cfg.inv_lambda) : 0.f;
    const float eta = wave_sum(w);
    if (slot >= 0) {
        if ((lane & 3) == 0 && ((lane >> 3) & 1) == 0) s_w[slot] = w;  // the leader lane of every slot (w = 0 when its sample does not exist)
    } else if ((lane & (LPS - 1)) == 0) s_w[lane / LPS] = w;  // non-leader lanes carry w = 0 and are not stored
    if (lane == 0) {
        rec[0] = beta;
        rec[1] = eta;
    }
    // this wavefront's own du stores must be visible to its other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    MPPI_BARRIER(5);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int nlive = K - k0 < SPW ? K - k0 : SPW;  // samples of this chunk that exist
    if (nlive == SPW && (K & 3) == 0) {  // aligned full chunk: 16-byte loads, the rows of up to four trips requested at once
        // (the du rows come back from L2 one round trip per trip of this loop otherwise - three of them in front of the
        // kernel's end for the panda's 140 rows)
        constexpr int kTrips = 4;
        float w[SPW];
#pragma unroll
        for (int q = 0; q < SPW; q++) w[q] = s_w[q];
        for (int j0 = lane; j0 < HN; j0 += kTrips * kWave) {
            float4 v[kTrips][SPW / 4];
#pragma unroll
            for (int t = 0; t < kTrips; t++) {
                const int j = j0 + t * kWave;
                if (j < HN) {
                    const float4 *row = reinterpret_cast<const float4 *>(du + (size_t)j * K + k0);
#pragma unroll
                    for (int q = 0; q < SPW / 4; q++) v[t][q] = row[q];
                }
            }
#pragma unroll
            for (int t = 0; t < kTrips; t++) {
                const int j = j0 + t * kWave;
                if (j < HN) {
                    float acc = 0.f;
#pragma unroll
                    for (int q = 0; q < SPW / 4; q++)
                        acc += v[t][q].x * w[4 * q] + v[t][q].y * w[4 * q + 1] + v[t][q].z * w[4 * q + 2] + v[t][q].w * w[4 * q + 3];
                    rec[2 + j] = acc;
                }
            }
        }
        return;
    }
    for (int j = lane; j < HN; j += kWave) {
        const float *row = du + (size_t)j * K + k0;
        float acc = 0.f;
        for (int q = 0; q < nlive; q++) acc += row[q] * s_w[q];
        rec[2 + j] = acc;
    }
}

// The record of a WORKGROUP of two wavefronts that own eight samples each (the octet layout of the contact-free rollout, LAY = 8):
// 16 consecutive samples -> ONE record, as many records per launch as the quad layout leaves (the combine kernel's time follows
// their number: 512 records cost it 22.7 us, 256 18.9 us).  beta and eta go through LDS (one barrier each); the rows are summed
// by the 128 threads as in quad_record<16>.  `slot`: sample slot 0..7 of this lane in its wavefront (oct_slot()).
__device__ __forceinline__ void oct_record2(CCfg &cfg, float s, bool live_leader, const float *__restrict__ du, int k0, float *__restrict__ rec, int slot) {
    constexpr int NT = 2 * kWave, SPG = 16;
    __shared__ float s_w[SPG], s_b[2], s_e[2];
    const int K = cfg.K, HN = cfg.H * cfg.nu;
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x >> 6;
    const bool fin = live_leader && isfinite(s);
    const float bw = wave_min(fin ? s : INFINITY);
    if (lane == 0) s_b[wv] = bw;
    MPPI_BARRIER(5);
    const float beta = fminf(s_b[0], s_b[1]);
    const float w = fin ? __expf(-(s - beta) * cfg.inv_lambda) : 0.f;
    const float ew = wave_sum(w);
    if (lane == 0) s_e[wv] = ew;
    if ((lane & 3) == 0 && ((lane >> 3) & 1) == 0) s_w[wv * 8 + slot] = w;  // the leader lane of every slot (w = 0: no such sample)
    // this workgroup's own du stores must be visible to all of its lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    MPPI_BARRIER(6);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (threadIdx.x == 0) {
        rec[0] = beta;
        rec[1] = s_e[0] + s_e[1];
    }
    const int nlive = K - k0 < SPG ? K - k0 : SPG;
    if (nlive == SPG && (K & 3) == 0) {  // aligned full group: 16-byte loads, both trips of a thread requested at once
        constexpr int kTrips = 2;
        float wq[SPG];
#pragma unroll
        for (int q = 0; q < SPG; q++) wq[q] = s_w[q];
        for (int j0 = threadIdx.x; j0 < HN; j0 += kTrips * NT) {
            float4 v[kTrips][SPG / 4];
#pragma unroll
            for (int t = 0; t < kTrips; t++) {
                const int j = j0 + t * NT;
                if (j < HN) {
                    const float4 *row = reinterpret_cast<const float4 *>(du + (size_t)j * K + k0);
#pragma unroll
                    for (int q = 0; q < SPG / 4; q++) v[t][q] = row[q];
                }
            }
#pragma unroll
            for (int t = 0; t < kTrips; t++) {
                const int j = j0 + t * NT;
                if (j < HN) {
                    float acc = 0.f;
#pragma unroll
                    for (int q = 0; q < SPG / 4; q++)
                        acc += v[t][q].x * wq[4 * q] + v[t][q].y * wq[4 * q + 1] + v[t][q].z * wq[4 * q + 2] + v[t][q].w * wq[4 * q + 3];
                    rec[2 + j] = acc;
                }
            }
        }
        return;
    }
    for (int j = threadIdx.x; j < HN; j += NT) {
        const float *row = du + (size_t)j * K + k0;
        float acc = 0.f;
        for (int q = 0; q < nlive; q++) acc += row[q] * s_w[q];
        rec[2 + j] = acc;
    }
}

// Second level of the record tree, still inside the rollout kernel.  The quad kernels deal their chunks so that the
// workgroups b with b % 8 == g - one XCD - own the contiguous chunks [g*n, (g+1)*n).  Every workgroup takes a ticket
// from a device-scope counter of its group after publishing its wave record; the LAST one of a group folds the group's
// n wave records (they sit in its own XCD's L2) into one record, in chunk order whichever workgroup happens to do it:
//   beta = min beta_r, eta = sum eta_r e^{-(beta_r-beta)/lambda}, N = sum N_r e^{-(beta_r-beta)/lambda}.
// The combine / update kernel - and the all-gather between GPUs - then deal with 8 records per GPU instead of K/16, and
// no separate "reduce to a shard record" launch exists on the fused path.  The counter re-arms itself.
constexpr int kFoldGroups = 8;
constexpr int kFoldTable = 1024;
// NW = wavefronts per workgroup: ONE ticket per workgroup (thread 0); with a helper wavefront (NW = 2) the ticket travels
// through LDS so that both wavefronts take the same way through the barriers below, and the helper does none of the work.
template <int NW = 1>
__device__ __forceinline__ void fold_group(CCfg &cfg, const float *__restrict__ partials, int first, int n, unsigned *__restrict__ ctr,
                                           float *__restrict__ out) {
    __shared__ float s_sc[kFoldTable];
    __shared__ unsigned s_ticket;
    const int lane = threadIdx.x & (kWave - 1);
    // Hand-off between workgroups (per-XCD L2s are not coherent, a CU's L1 is never refreshed by other CUs' stores):
    // producer = plain stores -> wait for them -> ONE lane's agent-scope release -> wait again (ROCm 7.2 may drop the
    // wait behind buffer_wbl2 when its scoreboard looks empty) -> relaxed agent-scope ticket; consumer = the last
    // ticket holder: ONE lane's agent-scope acquire -> barrier -> plain loads.  Placement-independent: only the speed
    // of the fold depends on the group really sharing an XCD.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MPPI_BARRIER(6);
    unsigned ticket = 0;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ticket = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (NW > 1) s_ticket = ticket;
    }
    if constexpr (NW > 1) {
        MPPI_BARRIER(7);
        ticket = s_ticket;
    } else {
        ticket = __shfl(ticket, 0, kWave);
    }
    if (ticket != (unsigned)(n - 1)) return;  // (workgroup-uniform)
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the next launch
    }
    MPPI_BARRIER(8);
    if constexpr (NW > 1)
        if (threadIdx.x >= kWave) {  // helper wavefront: the barrier of the fold below, none of its work
            MPPI_BARRIER(9);
            return;
        }
    const int HN = cfg.H * cfg.nu, RF = 2 + HN;
    const float *recs = partials + (size_t)first * RF;
    float b = INFINITY;
    for (int r = lane; r < n; r += kWave)
        if (recs[(size_t)r * RF + 1] > 0.f) b = fminf(b, recs[(size_t)r * RF]);
    b = wave_min(b);
    float e = 0.f;
    for (int r = lane; r < n; r += kWave) {
        const float er = recs[(size_t)r * RF + 1];
        const float sc = er > 0.f ? __expf(-(recs[(size_t)r * RF] - b) * cfg.inv_lambda) : 0.f;
        if (r < kFoldTable) s_sc[r] = sc;
        e += er * sc;
    }
    e = wave_sum(e);
    MPPI_BARRIER(9);
    if (lane == 0) {
        out[0] = b;
        out[1] = e;
    }
    const int nt = n < kFoldTable ? n : kFoldTable;
    for (int j = lane; j < HN; j += kWave) {
        const float *col = recs + 2 + j;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int r = 0;
        for (; r + 4 <= nt; r += 4) {  // four independent chains: the loads of a trip are in flight together
            a0 += s_sc[r] * col[(size_t)r * RF];
            a1 += s_sc[r + 1] * col[(size_t)(r + 1) * RF];
            a2 += s_sc[r + 2] * col[(size_t)(r + 2) * RF];
            a3 += s_sc[r + 3] * col[(size_t)(r + 3) * RF];
        }
        for (; r < nt; r++) a0 += s_sc[r] * col[(size_t)r * RF];
        for (; r < n; r++) {  // more records than the table holds (K > 131072 on one GPU): recompute the factor
            const float er = recs[(size_t)r * RF + 1];
            a0 += (er > 0.f ? __expf(-(recs[(size_t)r * RF] - b) * cfg.inv_lambda) : 0.f) * col[(size_t)r * RF];
        }
        out[2 + j] = (a0 + a1) + (a2 + a3);
    }
}
// chunk owned by workgroup b of nb (XCD-aware dealing, see k_rollout_quad) and the fold group it reports to
template <int NW = 1>
__device__ __forceinline__ void fold_after_record(CCfg &cfg, const float *__restrict__ partials, unsigned *__restrict__ fold_ctr,
                                                  float *__restrict__ fold_out) {
    if (fold_ctr == nullptr) return;
    const int nb = gridDim.x, RF = 2 + cfg.H * cfg.nu;
    if (nb % 16 == 0) {
        const int g = (int)(blockIdx.x % kFoldGroups), n = nb / kFoldGroups;
        fold_group<NW>(cfg, partials, g * n, n, fold_ctr + g, fold_out + (size_t)g * RF);
    } else {
        fold_group<NW>(cfg, partials, 0, nb, fold_ctr, fold_out);  // ragged grids (identity dealing): one group
    }
}

__global__ __launch_bounds__(kWave) void k_reduce(const DevCfg *__restrict__ cfg, const float *__restrict__ S,
                                                  const float *__restrict__ du, float *__restrict__ partials) {
    const int k = blockIdx.x * kWave + threadIdx.x;
    const bool live = k < cfg->K;
    wave_record(*(CCfg *)cfg, live ? S[k] : INFINITY, live, du, k, partials + (size_t)blockIdx.x * (2 + cfg->H * cfg->nu));
}

// the same records from 16 samples per wavefront (K/16 wavefronts instead of K/64: the per-row sums of quad_record instead of
// a 64-lane butterfly per row) - the record kernel of the generic mode was 24 us at K = 4096 with 64 wavefronts
__global__ __launch_bounds__(kWave) void k_reduce_quad(const DevCfg *__restrict__ cfg, const float *__restrict__ S,
                                                       const float *__restrict__ du, float *__restrict__ partials) {
    const int k0 = blockIdx.x * 16, k = k0 + (int)(threadIdx.x >> 2);
    const bool live = k < cfg->K;
    quad_record<16>(*(CCfg *)cfg, live ? S[k] : INFINITY, live && (threadIdx.x & 3) == 0, du, k0, partials + (size_t)blockIdx.x * (2 + cfg->H * cfg->nu));
}

// generic Objective mode, whole horizon at once (ABI 8): the host-side stage costs of all H*K env-steps come as ONE array
// c[t*K + k]; S_k += sum_t gamma^t c[t][k] and the per-wavefront records in the same launch - what `(c.view(H, K) * disc).sum(0)`
// (two torch kernels), mppi_sim_accumulate_cost, mppi_sim_finish and mppi_reduce did in five
__global__ __launch_bounds__(kWave) void k_horizon_reduce_quad(const DevCfg *__restrict__ cfg, const float *__restrict__ c, const float *__restrict__ ctrl,
                                                               float *__restrict__ S, const float *__restrict__ du, float *__restrict__ partials) {
    const int K = cfg->K, H = cfg->H;
    const int k0 = blockIdx.x * 16, k = k0 + (int)(threadIdx.x >> 2);
    const bool live = k < K;
    float s = INFINITY;
    if (live) {
        float acc = 0.f, disc = 1.f;
        for (int t = 0; t < H; t++) {
            acc = fmaf(disc, c[(size_t)t * K + k], acc);
            disc *= cfg->gamma;
        }
        s = S[k] + acc + ctrl[k];
        if ((threadIdx.x & 3) == 0) S[k] = s;
    }
    quad_record<16>(*(CCfg *)cfg, s, live && (threadIdx.x & 3) == 0, du, k0, partials + (size_t)blockIdx.x * (2 + H * cfg->nu));
}

template <class T>
__global__ __launch_bounds__(kWave) void k_rollout(const DevModel *__restrict__ m, const DevCfg *__restrict__ cfg,
                                                   const DevCost *__restrict__ cost, const float *__restrict__ x0_dof,
                                                   const float *__restrict__ x0_root, const float *__restrict__ U,
                                                   const float *__restrict__ eps, const float *__restrict__ prior,
                                                   float *__restrict__ du, float *__restrict__ S, float *__restrict__ viz,
                                                   float *__restrict__ partials) {
    const int k = blockIdx.x * kWave + threadIdx.x;
    const bool live = k < cfg->K;
    float s = INFINITY;
    if (live) {
        s = rollout_sample<T>(*(CModel *)m, *(CCfg *)cfg, *(CCost *)cost, x0_dof, x0_root, U, eps, prior, du, viz, k);
        S[k] = s;
    }
    // fused tail: this wave's partial record (its own du writes are visible to its own lanes)
    wave_record(*(CCfg *)cfg, s, live, du, k, partials + (size_t)blockIdx.x * (2 + cfg->H * cfg->nu));
}

// Quad-parallel rollout (mppi_quad.hpp): 4 lanes per sample, 16 samples per wavefront.
// LAY = 8: the OCTET layout of the articulated-body solve (mppi_oct.hpp) - 8 lanes per sample, the angular half of every spatial
// quantity in one quad and the linear half in the other, 8 samples per wavefront (two per 16-lane row): the same step / rollout
// code around a solve with ~15 % fewer issue slots (tools/exp/oct_aba_proto.hip), K/8 wavefronts.
template <class T, bool DUMP = false, int LAY = 4>
// (No amdgpu_waves_per_eu(1, 1) here although the kernel runs one wavefront per SIMD by construction: measured, round 3, the
// attribute makes this kernel 20 % SLOWER (0.1175 -> 0.1415 ms, same instruction counts, 256 + 1 registers instead of 256 and
// a few spilled loop invariants) - the register file above 256 is not free for a wavefront that could live without it.)
__global__ __launch_bounds__(LAY == 8 ? 2 * kWave : kWave) void k_rollout_quad(const DevModel *__restrict__ m, const DevCfg *__restrict__ cfg,
                                                        const DevCost *__restrict__ cost, const float *__restrict__ x0_dof,
                                                        const float *__restrict__ x0_root, const float *__restrict__ U,
                                                        const float *__restrict__ eps, const float *__restrict__ prior,
                                                        float *__restrict__ du, float *__restrict__ S, float *__restrict__ viz,
                                                        float *__restrict__ partials, unsigned *__restrict__ fold_ctr, float *__restrict__ fold_out,
                                                        unsigned long long *__restrict__ wave_clk, float *__restrict__ traj = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)  // (the host pass sees the 4-float emulation type of mppi_quad.hpp)
    const unsigned long long clk0 = wave_clk != nullptr ? wall_clock64() : 0ull;
    // Stage the robot model (header + body + link blocks, ~4 KB) in LDS once per wavefront: constants are then
    // fetched with in-order ds_read_b128 broadcasts into VGPRs - no SMEM round trip (s_waitcnt lgkmcnt(0) on
    // every block), no SGPR spills, no constant-bus moves.
    // (LAY = 8: a workgroup is TWO wavefronts of eight samples each - 16 consecutive samples, one record, one staged model)
    constexpr int NT = LAY == 8 ? 2 * kWave : kWave;
    constexpr int kModelBytes = (int)((offsetof(DevModel, fr) + 15) / 16 * 16);
    __shared__ __attribute__((aligned(64))) uint4 s_model[kModelBytes / 16];
    // (the model's loads are all requested first and written to LDS last: their round trip to memory runs under the
    // staging of the step constants, which has round trips of its own)
    constexpr int kModelTrips = (kModelBytes / 16 + NT - 1) / NT;
    uint4 mv[kModelTrips];
#pragma unroll
    for (int it = 0; it < kModelTrips; it++) {
        const int i = (int)threadIdx.x + it * NT;
        mv[it] = reinterpret_cast<const uint4 *>(m)[i < kModelBytes / 16 ? i : 0];  // (unconditional: the array stays in registers)
    }
    // ... and the step loop's own constants (control limits, nominal rows, cost target and weights)
    __shared__ __attribute__((aligned(64))) float s_step[sizeof(StepConsts) / sizeof(float)];
    {
        const int n = step_const_count(*(CCfg *)cfg);
        for (int j = threadIdx.x; j < n; j += NT) s_step[j] = step_const_entry(*(CCfg *)cfg, *(CCost *)cost, x0_root, U, j);
    }
#pragma unroll
    for (int it = 0; it < kModelTrips; it++) {
        const int i = (int)threadIdx.x + it * NT;
        if (i < kModelBytes / 16) s_model[i] = mv[it];
    }
    // octet layout: the linear lanes read the bodies' inertia blocks from a copy whose inertia tensors and 1/m are zero
    // (mppi_oct.hpp oct_lin_view); lane i stages body i
    __shared__ __attribute__((aligned(256))) uint4 s_lin_raw[(LAY == 8 ? oct_lin_raw_bytes(T::NB) : 16) / 16];
    LModel &lm = *(LModel *)s_model;
    LStep &sc = *(LStep *)s_step;
    MPPI_LDS_AS DevBody *s_lin = nullptr;
    if constexpr (LAY == 8) {
        s_lin = oct_lin_place((MPPI_LDS_AS void *)s_lin_raw, &lm.b[0]);   // (bank-disjoint from the model's own blocks)
        if ((int)threadIdx.x < T::NB) {
            DevBody b = ((const DevModel *)m)->b[threadIdx.x];
            b.k1 = oct_lin_view(b.k1);
            s_lin[threadIdx.x] = b;
        }
    }
    __syncthreads();
    // XCD-aware chunk mapping: a wavefront owns 16 consecutive samples = 64 B of every sample-minor row, i.e.
    // half a 128-B line.  Workgroup b runs on XCD b % 8 and the XCD L2s are private, so with the identity
    // mapping the two halves of each line are fetched by two different L2s (measured: 2x the algorithmic
    // read traffic).  Chunks are therefore dealt so that chunks 2i and 2i+1 land on the same XCD.
    const int nb = gridDim.x;
    const int chunk = (nb % 16 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;   // 16 samples either way
    const int k = chunk * 16 + (LAY == 8 ? (int)(threadIdx.x >> 6) * 8 + oct_slot() : (int)(threadIdx.x >> 2));
    const int lane4 = threadIdx.x & 3;
    const bool live = k < cfg->K;        // the lanes of a sample share k
    const bool leader = lane4 == 0 && (LAY == 4 || oct_half() == 0);
    float s = INFINITY;
    if (live) {
        // one wave-uniform branch picks the instruction stream specialised for an all-revolute tree
        if constexpr (LAY == 8) {
            const OctAba ab{oct_bodies(&lm.b[0], s_lin), oct_lane()};
            if (((CModel *)m)->all_revolute) s = quad_rollout<T, 0, DUMP>(lm, *(CCfg *)cfg, *(CCost *)cost, sc, x0_dof, x0_root, eps, prior, du, viz, k, leader, quad_row(), lane4 < 3, traj, ab);
            else s = quad_rollout<T, -1, DUMP>(lm, *(CCfg *)cfg, *(CCost *)cost, sc, x0_dof, x0_root, eps, prior, du, viz, k, leader, quad_row(), lane4 < 3, traj, ab);
        } else {
            if (((CModel *)m)->all_revolute) s = quad_rollout<T, 0, DUMP>(lm, *(CCfg *)cfg, *(CCost *)cost, sc, x0_dof, x0_root, eps, prior, du, viz, k, leader, quad_row(), lane4 < 3, traj);
            else s = quad_rollout<T, -1, DUMP>(lm, *(CCfg *)cfg, *(CCost *)cost, sc, x0_dof, x0_root, eps, prior, du, viz, k, leader, quad_row(), lane4 < 3, traj);
        }
        if (leader) S[k] = s;
    }
    if constexpr (LAY == 8) {
        oct_record2(*(CCfg *)cfg, s, live && leader, du, chunk * 16, partials + (size_t)chunk * (2 + cfg->H * cfg->nu), oct_slot());
        fold_after_record<2>(*(CCfg *)cfg, partials, fold_ctr, fold_out);
    } else {
        quad_record(*(CCfg *)cfg, s, live && leader, du, chunk * 16, partials + (size_t)chunk * (2 + cfg->H * cfg->nu));
        fold_after_record(*(CCfg *)cfg, partials, fold_ctr, fold_out);
    }
    if (wave_clk != nullptr && threadIdx.x == 0) {  // instrumentation (mppi_set_wave_clock): this wavefront's residency
        wave_clk[2 * chunk] = clk0;
        wave_clk[2 * chunk + 1] = wall_clock64();
    }
#endif
}

// Combine n records (1024 threads): beta = min, eta and N rescaled by e^{-(beta_r-beta)/lambda}.
// mode 0: write the combined record to `out`; mode 1: U += N/eta, action = U[0], shift U, append u_init.
// The row sums over the records are split over up to eight groups of H*nu threads (record r belongs to group r mod G).
constexpr int kCombineThreads = 1024;
constexpr int kCombineGroups = 8;
// the fused tail also steps the K = 1 world on one quad: 512 threads leave that quad 256 registers (at 1024 threads the
// step spilled 108 B to scratch)
constexpr int kCombineWorldThreads = 512;
template <int NT>
__device__ __forceinline__ void combine_update(CCfg &cfg, const float *__restrict__ recs, int nrec, int mode, float *__restrict__ out,
                                               float *__restrict__ U, float *__restrict__ action, float *__restrict__ beta_eta, float *s_act,
                                               const float *__restrict__ filt) {
    __shared__ float s_red[NT];
    __shared__ float s_part[kCombineGroups][MPPI_MAX_H * MPPI_MAX_NU];
    constexpr int kMaxScale = 4096;
    __shared__ float s_scale[kMaxScale];
    const int HN = cfg.H * cfg.nu, RF = 2 + HN, nu = cfg.nu;
    const int tid = threadIdx.x;
    // block reductions: wave shuffles, then one 16-entry pass (two barriers instead of ten per reduction)
    const int wid = tid >> 6, lane = tid & 63;
    // every load that does not depend on a reduction is issued up front: this kernel is a chain of memory round trips
    // (records written by the other XCDs come from the fabric, ~2 us each), not bandwidth
    const float U_old = (mode != 0 && tid < HN) ? U[tid] : 0.f;
    const bool mine = tid < nrec;  // the first NT records stay in registers between the two passes
    const float er0 = mine ? recs[(size_t)tid * RF + 1] : 0.f, br0 = mine ? recs[(size_t)tid * RF] : 0.f;
    float b = er0 > 0.f ? br0 : INFINITY;
    for (int r = tid + NT; r < nrec; r += NT)
        if (recs[(size_t)r * RF + 1] > 0.f) b = fminf(b, recs[(size_t)r * RF]);
    b = wave_min(b);
    if (lane == 0) s_red[wid] = b;
    __syncthreads();
    float beta = s_red[lane & (NT / kWave - 1)];
#pragma unroll
    for (int o = NT / kWave / 2; o > 0; o >>= 1) beta = fminf(beta, __shfl_xor(beta, o, kWave));
    __syncthreads();
    float e = 0.f;
    if (mine) {
        const float sc = er0 > 0.f ? __expf(-(br0 - beta) * cfg.inv_lambda) : 0.f;
        s_scale[tid] = sc;
        e = er0 * sc;
    }
    for (int r = tid + NT; r < nrec; r += NT) {
        const float er = recs[(size_t)r * RF + 1];
        const float sc = er > 0.f ? __expf(-(recs[(size_t)r * RF] - beta) * cfg.inv_lambda) : 0.f;
        if (r < kMaxScale) s_scale[r] = sc;
        e += er * sc;
    }
    e = wave_sum(e);
    if (lane == 0) s_red[wid] = e;
    __syncthreads();
    float eta = s_red[lane & (NT / kWave - 1)];
#pragma unroll
    for (int o = NT / kWave / 2; o > 0; o >>= 1) eta += __shfl_xor(eta, o, kWave);
    // N[j] = sum_r scale_r * recs[r][2 + j]: the block is cut into G groups of HN threads (one thread per row j, ONE pass
    // whatever HN is); group g takes the records r = g, g + G, ...; the loop is unrolled so that sixteen independent
    // loads are in flight per thread instead of a dependent chain
    // Wide variant (every record in the weight table): a thread takes FOUR adjacent rows, so a
    // quarter of the threads cover the rows, four times as many groups share the records and the ~nrec/G 16-byte loads of a
    // thread are all in flight at once - one round trip to memory instead of one per sixteen records.
    constexpr int kPartFloats = kCombineGroups * MPPI_MAX_H * MPPI_MAX_NU;
    constexpr int kWideLoads = NT >= 1024 ? 16 : 24;  // (register budget: 128 per thread in a 1024-thread block)
    const int Q = (HN + 3) >> 2;  // (the last quad of a row count that is no multiple of four is loaded float by float)
    int Gw = Q > 0 ? NT / Q : 0;
    if (Gw > 32) Gw = 32;
    if (Gw * HN > kPartFloats) Gw = kPartFloats / HN;
    const bool wide = Q <= NT && nrec <= kMaxScale && Gw >= 2;
    float *s_flat = &s_part[0][0];
    int G = HN <= NT ? (NT / HN < kCombineGroups ? NT / HN : kCombineGroups) : 1;
    if (wide) {
        G = Gw;
        const int g = tid / Q, cq = tid - g * Q;
        if (g < G) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 v[kWideLoads];
            const int nvalid = HN - 4 * cq;  // >= 4 except in the last quad of a ragged row count
            auto load4 = [&](int r, float4 &w) {
                const float *src = recs + (size_t)r * RF + 2 + 4 * cq;
                if (nvalid >= 4) __builtin_memcpy(&w, src, 16);
                else {
                    w.x = src[0];
                    w.y = nvalid > 1 ? src[1] : 0.f;
                    w.z = nvalid > 2 ? src[2] : 0.f;
                    w.w = 0.f;
                }
            };
#pragma unroll
            for (int i = 0; i < kWideLoads; i++) {
                const int r = g + i * G;
                if (r < nrec) load4(r, v[i]);
            }
#pragma unroll
            for (int i = 0; i < kWideLoads; i++) {
                const int r = g + i * G;
                if (r < nrec) {
                    const float sc = s_scale[r];
                    acc.x += v[i].x * sc; acc.y += v[i].y * sc; acc.z += v[i].z * sc; acc.w += v[i].w * sc;
                }
            }
#pragma unroll 8
            for (int r = g + kWideLoads * G; r < nrec; r += G) {
                float4 w;
                load4(r, w);
                const float sc = s_scale[r];
                acc.x += w.x * sc; acc.y += w.y * sc; acc.z += w.z * sc; acc.w += w.w * sc;
            }
            float *o = s_flat + g * HN + 4 * cq;
            o[0] = acc.x;
            if (nvalid > 1) o[1] = acc.y;
            if (nvalid > 2) o[2] = acc.z;
            if (nvalid > 3) o[3] = acc.w;
        }
    } else {
        const int g = tid / HN, j0 = tid - g * HN;
        if (g < G)
            for (int j = j0; j < HN; j += NT) {  // (more than one trip only if HN > 1024, then G = 1)
                float N0 = 0.f;
#pragma unroll 16
                for (int r = g; r < nrec; r += G) {
                    float sc;
                    if (r < kMaxScale) sc = s_scale[r];
                    else {  // more records than the LDS table holds (K > 65536 on one GPU): recompute the rescaling factor
                        const float er = recs[(size_t)r * RF + 1];
                        sc = er > 0.f ? __expf(-(recs[(size_t)r * RF] - beta) * cfg.inv_lambda) : 0.f;
                    }
                    N0 += recs[(size_t)r * RF + 2 + j] * sc;
                }
                s_part[g][j] = N0;
            }
    }
    __syncthreads();
    // the updated nominal goes to the weight table's storage (consumed above; HN <= 1024 floats each for U and F U)
    float *s_U = s_scale;
    float Unew = 0.f;
    const int gstride = wide ? HN : MPPI_MAX_H * MPPI_MAX_NU;
    if (tid < 256)
        for (int j = tid; j < HN; j += 256) {
            float N = 0.f;
            for (int gg = 0; gg < G; gg++) N += s_flat[gg * gstride + j];
            if (mode == 0) out[2 + j] = N;
            else {
                Unew = (j == tid ? U_old : U[j]) + (eta > 0.f ? N / eta : 0.f);
                s_scale[j] = Unew;
            }
        }
    if (mode == 0) {
        if (tid == 0) {
            out[0] = beta;
            out[1] = eta;
        }
        return;
    }
    __syncthreads();
    if (filt != nullptr) {  // filter_u: U <- F U over the horizon, per control dimension
        const int H = cfg.H;
        for (int j = tid; j < HN; j += NT) {
            const int t = j / nu, c = j - t * nu;
            float acc = 0.f;
            for (int s2 = 0; s2 < H; s2++) acc += filt[t * H + s2] * s_scale[s2 * nu + c];
            s_scale[MPPI_MAX_H * MPPI_MAX_NU + j] = acc;
        }
        __syncthreads();
        s_U = s_scale + MPPI_MAX_H * MPPI_MAX_NU;
    }
    if (tid < nu) {
        action[tid] = s_U[tid];
        if (s_act != nullptr) s_act[tid] = s_U[tid];
        float *mirror = cfg.action_mirror;
        if (mirror != nullptr) mirror[tid] = s_U[tid];
    }
    // publish: the action stores above come from this same wavefront (nu <= 64), so the fence of lane 0 covers them
    if (tid == 0 && cfg.seq_host != nullptr) {
        __threadfence_system();
        unsigned *sd = cfg.seq_dev;
        const unsigned sq = *sd + 1u;
        *sd = sq;
        __hip_atomic_store(cfg.seq_host, sq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tid == 0) {
        beta_eta[0] = beta;
        beta_eta[1] = eta;
    }
    for (int j = tid; j < HN; j += NT) U[j] = (j + nu < HN) ? s_U[j + nu] : cfg.u_init;  // shift, append u_init
}

__global__ __launch_bounds__(kCombineThreads) void k_combine(const DevCfg *__restrict__ cfg, const float *__restrict__ recs, int nrec, int mode,
                                                             float *__restrict__ out, float *__restrict__ U, float *__restrict__ action,
                                                             float *__restrict__ beta_eta, const float *__restrict__ filt) {
    combine_update<kCombineThreads>(*(CCfg *)cfg, recs, nrec, mode, out, U, action, beta_eta, nullptr, filt);
}


// ---- direct exchange of the shard records between the GPUs of a node (the 1-hop mailbox all-gather) ----------
// Every rank owns an INBOX in device memory that its peers can write (same process: plain device pointers, other processes:
// hipIpc handles, other GPUs: xGMI peer access): n flags (one 64-byte line each) and 2 x n slots of `nrec` records.  Per
// control iteration a rank PUBLISHES its records into slot [seq & 1][rank] of every inbox (its own included), releases, and
// stores the iteration's sequence number into flag [rank] of every inbox; then it WAITS until the n flags of its own inbox have
// reached the sequence number, acquires, and copies the n x nrec records of that parity into a buffer with a fixed address -
// what the combine / update kernels read (fixed pointers: the iteration can live in a captured graph).  Two slots are enough:
// a rank cannot publish iteration i+2 before every peer has published i+1, which a peer does only after it has consumed i.
// The spin is bounded (status word 1 = timed out: a peer never published); release / acquire are system-scope, so the same
// code serves one device (tests: N contexts on N streams) and xGMI peers.
constexpr int kMailboxFlagStride = 16;  // uints: one flag per 64-byte line
struct MailboxHeader {                  // layout of an inbox: flags [n][16] uint, then records [2][n][nrec][RF] float
    static size_t bytes(int n, int nrec, int RF) { return sizeof(unsigned) * kMailboxFlagStride * (size_t)n + sizeof(float) * 2 * (size_t)n * nrec * RF; }
};
__device__ __forceinline__ float *mailbox_slot(void *inbox, int n, int nrec, int RF, int parity, int rank) {
    return reinterpret_cast<float *>(reinterpret_cast<unsigned *>(inbox) + (size_t)kMailboxFlagStride * n) + ((size_t)parity * n + rank) * nrec * RF;
}
__device__ __forceinline__ void mailbox_publish(const float *__restrict__ own, int nrec, int RF, int rank, int n, void *const *__restrict__ peers,
                                                unsigned *__restrict__ seq_ctr) {
    const unsigned seq = *seq_ctr + 1u;
    const int parity = (int)(seq & 1u), len = nrec * RF;
    for (int p = 0; p < n; p++) {
        float *dst = mailbox_slot(peers[p], n, nrec, RF, parity, rank);
        for (int j = threadIdx.x; j < len; j += blockDim.x) dst[j] = own[j];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the records are visible before the flags
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int p = 0; p < n; p++)
            __hip_atomic_store(reinterpret_cast<unsigned *>(peers[p]) + (size_t)kMailboxFlagStride * rank, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(seq_ctr, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void mailbox_wait(void *__restrict__ inbox, int nrec, int RF, int n, const unsigned *__restrict__ seq_ctr,
                                             float *__restrict__ gathered, unsigned *__restrict__ status, unsigned long long max_ticks) {
    __shared__ int s_late;
    __shared__ unsigned s_seq;
    if (threadIdx.x == 0) {
        s_late = 0;
        s_seq = __hip_atomic_load(seq_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (stored by the publish ahead of this wait)
    }
    __syncthreads();
    const unsigned seq = s_seq;
    bool late = false;  // thread r polls rank r's flag
    if ((int)threadIdx.x < n) {
        const unsigned *flag = reinterpret_cast<const unsigned *>(inbox) + (size_t)kMailboxFlagStride * threadIdx.x;
        const unsigned long long t0 = wall_clock64();
        // (sequence numbers only grow; the comparison survives the 2^32 wrap)
        while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > max_ticks) {
                late = true;
                s_late = 1;
                break;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        if (s_late) *status = 1u;
    }
    __syncthreads();
    const float *src = mailbox_slot(inbox, n, nrec, RF, (int)(seq & 1u), 0);
    const int len = n * nrec * RF;
    for (int j = threadIdx.x; j < len; j += blockDim.x) gathered[j] = __builtin_nontemporal_load(src + j);
    // a rank that never published this iteration: whatever its slot holds (an older iteration's records, or half-written ones) must
    // not reach the update - its records become neutral (eta = 0: the combine skips them), the status word says what happened
    if (s_late) {
        __syncthreads();
        if (late)
            for (int i = 0; i < nrec; i++) gathered[((size_t)threadIdx.x * nrec + i) * RF + 1] = 0.f;
    }
}
__global__ __launch_bounds__(256) void k_mailbox_publish(const float *__restrict__ own, int nrec, int RF, int rank, int n, void *const *__restrict__ peers,
                                                         unsigned *__restrict__ seq_ctr) {
    mailbox_publish(own, nrec, RF, rank, n, peers, seq_ctr);
}
__global__ __launch_bounds__(256) void k_mailbox_wait(void *__restrict__ inbox, int nrec, int RF, int n, const unsigned *__restrict__ seq_ctr,
                                                      float *__restrict__ gathered, unsigned *__restrict__ status, unsigned long long max_ticks) {
    mailbox_wait(inbox, nrec, RF, n, seq_ctr, gathered, status, max_ticks);
}
// both halves in one launch (mppi_exchange: one kernel boundary less per control iteration)
__global__ __launch_bounds__(256) void k_mailbox_exchange(const float *__restrict__ own, int nrec, int RF, int rank, int n, void *const *__restrict__ peers,
                                                          unsigned *__restrict__ seq_ctr, void *__restrict__ inbox, float *__restrict__ gathered,
                                                          unsigned *__restrict__ status, unsigned long long max_ticks) {
    mailbox_publish(own, nrec, RF, rank, n, peers, seq_ctr);
    __syncthreads();
    mailbox_wait(inbox, nrec, RF, n, seq_ctr, gathered, status, max_ticks);
}

// ... and, for the rollouts that leave per-wavefront records (contact-free kernels: the in-kernel fold is off there), the
// reduction of those records to the ONE shard record in front of it, in the same launch: one kernel boundary less per
// sharded iteration (mppi_reduce + mppi_exchange were two launches)
__global__ __launch_bounds__(kCombineThreads) void k_reduce_exchange(const DevCfg *__restrict__ cfg, const float *__restrict__ recs, int nrec_in,
                                                                     float *__restrict__ own, int RF, int rank, int n, void *const *__restrict__ peers,
                                                                     unsigned *__restrict__ seq_ctr, void *__restrict__ inbox, float *__restrict__ gathered,
                                                                     unsigned *__restrict__ status, unsigned long long max_ticks) {
    combine_update<kCombineThreads>(*(CCfg *)cfg, recs, nrec_in, 0, own, nullptr, nullptr, nullptr, nullptr, nullptr);
    __syncthreads();  // (the shard record is in memory and visible to the whole workgroup)
    mailbox_publish(own, 1, RF, rank, n, peers, seq_ctr);
    __syncthreads();
    mailbox_wait(inbox, 1, RF, n, seq_ctr, gathered, status, max_ticks);
}

// Closed-loop tail in ONE launch: combine + nominal update, then the K = 1 world is stepped with the new
// action by one quad of the same workgroup and its state becomes the planner's next x0
// (replaces k_combine + k_sim_step + k_state_from_world; fixed-base contact-free scenes only).
// With a mailbox (mb.inbox != null; sharded runs of the contact-free scenes, mppi_exchange_update_step_world) the same launch
// starts with this shard's part of the exchange: the per-wavefront records of the rollout are reduced to the ONE shard record,
// published into every rank's inbox, the ranks' records awaited - and the combine then reads the gathered records.  The
// sharded control iteration is two launches like the unsharded one.
struct MailboxArgs {
    const float *wave_recs = nullptr;  // this shard's per-wavefront records and their number
    int n_wave = 0;
    float *own = nullptr;              // [RF] the shard record
    int RF = 0, rank = 0, n = 0;
    void *const *peers = nullptr;
    unsigned *seq_ctr = nullptr;
    void *inbox = nullptr;
    float *gathered = nullptr;
    unsigned *status = nullptr;
    unsigned long long max_ticks = 0;
};
template <class T>
__global__ __launch_bounds__(kCombineWorldThreads) void k_combine_world(const DevCfg *__restrict__ cfg, const float *__restrict__ recs, int nrec,
                                                                   float *__restrict__ U, float *__restrict__ action, float *__restrict__ beta_eta,
                                                                   const DevModel *__restrict__ wm, const float *__restrict__ w_root,
                                                                   float *__restrict__ wq, float *__restrict__ wqd, float *__restrict__ x0_dof,
                                                                   const float *__restrict__ filt, MailboxArgs mb) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ float s_act[MPPI_MAX_NU];
    // The world's model goes to LDS as in the rollout (in-order ds_read_b128 broadcasts instead of cold scalar-cache
    // misses in the step's dependency chain).  The copy is made by the upper half of the block, whose wavefronts have no
    // record of their own to load in the first pass of the combine, so it costs the critical path nothing.
    constexpr int kModelBytes = (int)((offsetof(DevModel, fr) + 15) / 16 * 16);
    __shared__ __attribute__((aligned(64))) uint4 s_model[kModelBytes / 16];
    if (threadIdx.x >= kCombineWorldThreads / 2)
        for (int i = threadIdx.x - kCombineWorldThreads / 2; i < kModelBytes / 16; i += kCombineWorldThreads / 2)
            s_model[i] = reinterpret_cast<const uint4 *>(wm)[i];
    if (mb.inbox != nullptr) {
        combine_update<kCombineWorldThreads>(*(CCfg *)cfg, mb.wave_recs, mb.n_wave, 0, mb.own, nullptr, nullptr, nullptr, nullptr, nullptr);
        __syncthreads();  // (the shard record is in memory and visible to the whole workgroup)
        mailbox_publish(mb.own, 1, mb.RF, mb.rank, mb.n, mb.peers, mb.seq_ctr);
        __syncthreads();
        mailbox_wait(mb.inbox, 1, mb.RF, mb.n, mb.seq_ctr, mb.gathered, mb.status, mb.max_ticks);
        __syncthreads();
        recs = mb.gathered;
        nrec = mb.n;
    }
    combine_update<kCombineWorldThreads>(*(CCfg *)cfg, recs, nrec, 1, nullptr, U, action, beta_eta, s_act, filt);
    __syncthreads();
    if (threadIdx.x < 4) {
        constexpr int NB = T::NB;
        LModel &M = *(LModel *)s_model;
        QF q[NB ? NB : 1], qd[NB ? NB : 1], target[NB ? NB : 1];
        static_for<0, NB>([&](auto ic) {
            constexpr int i = ic;
            q[i] = wq[i];
            qd[i] = wqd[i];
            const CmdBlock b = load_block<CmdBlock>(M.b[i].cmd);
            float tg = 0.f;
#pragma unroll
            for (int c = 0; c < kMaxNu; c++) tg += b.v[c] * (c < M.nu ? s_act[c] : 0.f);
            target[i] = tg;
        });
        QPose<T> P;
        quad_base<T>(M, w_root, P);
        quad_fk<T>(M, q, P);
        quad_step<T>(M, P, q, qd, target);
        if (threadIdx.x == 0)
            static_for<0, NB>([&](auto ic) {
                constexpr int i = ic;
                wq[i] = q[i];
                wqd[i] = qd[i];
                x0_dof[2 * i] = q[i];
                x0_dof[2 * i + 1] = qd[i];
            });
    }
#endif
}

// ---- contact scenes (floating base, free bodies, penalty contact): per-lane working set in LDS ----
template <class T>
__global__ __launch_bounds__(kWave) void k_rollout_scene(const DevModel *__restrict__ m, const DevCfg *__restrict__ cfg,
                                                         const DevCost *__restrict__ cost, const float *__restrict__ x0_dof,
                                                         const float *__restrict__ x0_root, const float *__restrict__ U,
                                                         const float *__restrict__ eps, const float *__restrict__ prior,
                                                         float *__restrict__ du, float *__restrict__ S, float *__restrict__ viz,
                                                         float *__restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int k = blockIdx.x * kWave + threadIdx.x;
    const bool live = k < cfg->K;
    LMem L{lds + threadIdx.x, kWave};  // lane-minor rows: conflict-free ds_read/ds_write
    // (records of the light bodies' pairs, mppi_scene.hpp "light bodies": a per-lane array here - 64 rows of a large scene fill the LDS)
    float light_rec[kLightFloats];
    L.lp = light_rec;
    L.lstride = 1;
    if (((CModel *)m)->n_light_pairs != 0) light_region_reset(L);
    // start state in rollout coordinates (relative to the robot's start position, mppi_scene.hpp root_relative)
    __shared__ float s_root[13 * kMaxActors];
    root_origin(*(CModel *)m, x0_root, L.ox, L.oy);
    for (int j = threadIdx.x; j < 13 * m->n_actors; j += kWave) s_root[j] = root_relative_entry(x0_root, j, L.ox, L.oy);
    __syncthreads();
    float s = INFINITY;
    if (live) {
        s = rollout_scene<T>(*(CModel *)m, *(CModel *)m, *(CCfg *)cfg, *(CCost *)cost, x0_dof, s_root, U, eps, prior, du, viz, k, L);
        S[k] = s;
    }
    wave_record(*(CCfg *)cfg, s, live, du, k, partials + (size_t)blockIdx.x * (2 + cfg->H * cfg->nu));
}

// Contact scene, LPS = 4 or 8 lanes per sample.  The lanes of a sample replicate its state arithmetic (idle lanes cost
// nothing - at K=8192 one lane per sample is only 128 wavefronts on 1024 SIMDs), run the robot algebra in the quad layout
// (both quads of an octet alike) and deal the contact work over all LPS lanes (mppi_scene.hpp: kSplitQuad / kSplitOct).
// The sample's LDS rows are shared by its lanes: element i of the wave's s-th sample lives at lds[i*SPW + s]
// (SPW = 64 / LPS samples per wavefront; LPS-lane broadcast reads).
#if defined(MPPI_SCENE_WAVES_PER_EU)  // experiment builds (MPPI_BUILD_VARIANT=w2): register budget for N resident wavefronts per SIMD
#define MPPI_SCENE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(MPPI_SCENE_WAVES_PER_EU, MPPI_SCENE_WAVES_PER_EU)))
#else
#define MPPI_SCENE_OCCUPANCY
#endif
// OSOLVE: the articulated-body solve in the octet layout too (mppi_scene_oct.hpp: fixed-base trees of more than four bodies)
template <class T, int LPS, int NW = 1, bool DUMP = false, bool OSOLVE = false>
__global__ __launch_bounds__(kWave * NW) __attribute__((amdgpu_waves_per_eu(NW))) MPPI_SCENE_OCCUPANCY void k_rollout_scene_quad(const DevModel *__restrict__ m, const DevCfg *__restrict__ cfg,
                                                              const DevCost *__restrict__ cost, const float *__restrict__ x0_dof,
                                                              const float *__restrict__ x0_root, const float *__restrict__ U,
                                                              const float *__restrict__ eps, const float *__restrict__ prior,
                                                              float *__restrict__ du, float *__restrict__ S, float *__restrict__ viz,
                                                              float *__restrict__ partials, unsigned *__restrict__ fold_ctr, float *__restrict__ fold_out,
                                                              unsigned long long *__restrict__ wave_clk, float *__restrict__ traj = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(LPS == 4 || LPS == 8, "4 or 8 lanes per sample");
    static_assert(NW == 1 || (NW == 2 && LPS == 8 && T::NB <= 4), "helper wavefront: octet layout of the short trees only");
    constexpr int SPW = kWave / LPS;
    static_assert(!OSOLVE || (LPS == 8 && NW == 1 && T::NB > 4), "octet solve: octet kernel of the longer trees");
    constexpr int kSplit = NW == 2 ? kSplitOctPair : (LPS == 8 ? (OSOLVE ? kSplitOctSolve : kSplitOct) : kSplitQuad);
    const unsigned long long clk0 = wave_clk != nullptr ? wall_clock64() : 0ull;
#if defined(MPPI_SECTION_CLOCKS)
    if (threadIdx.x <= kSections) section_counters()[threadIdx.x] = threadIdx.x == kSections ? __builtin_readcyclecounter() : 0ull;
    __syncthreads();
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // The robot part of the model (header, bodies, links: 4.3 KB) is staged in LDS for the quad-layout kinematics and
    // articulated-body solve; shapes, pairs and free bodies stay behind the scalar cache (staging the WHOLE model was
    // measured slower: the contact loop's constants then occupy VGPRs of code that already spills)
    constexpr int kModelBytes = (int)((offsetof(DevModel, sh) + 15) / 16 * 16);  // header, bodies, links, free bodies
    __shared__ __attribute__((aligned(64))) uint4 s_model[kModelBytes / 16];
    for (int i = threadIdx.x; i < kModelBytes / 16; i += kWave * NW) s_model[i] = reinterpret_cast<const uint4 *>(m)[i];
    __syncthreads();
    LModel &lm = *(LModel *)s_model;
    const int nb = gridDim.x;  // XCD-aware chunk mapping as in k_rollout_quad (here 128 / (4 SPW) chunks share a line)
    const int chunk = (nb % 16 == 0) ? (int)(blockIdx.x % 8) * (nb / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
#if defined(MPPI_HWID_DEBUG)
    unsigned long long hwdbg = 0;
#endif
    // (NW == 2: the wavefront index decides who owns the sample state and who helps.  Measured placement on MI355X
    // (tools/exp/hwid_dump.py): the two wavefronts of the four workgroups of a CU go to SIMDs 1,3 | 3,0 | 0,2 | 2,1 - every SIMD
    // hosts one owner and one helper, which is what the solve phase, where helpers only wait, wants.)
#if defined(MPPI_HWID_DEBUG)
    if constexpr (NW == 2) {
        __shared__ unsigned s_hw[2];
        if (lane == 0) s_hw[wave] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SE [15:13]
        __syncthreads();
        hwdbg = (unsigned long long)s_hw[0] | ((unsigned long long)s_hw[1] << 32);
    }
#endif
    // lanes of a sample: a quad (LPS = 4), or two quads 8 lanes apart inside a 16-lane row (LPS = 8: the octet lane map of
    // mppi_oct.hpp - row_ror:8 reaches the partner quad); `sub` numbers them 0 .. LPS - 1
    const int slot = LPS == 8 ? oct_slot() : lane / LPS;
    const int k = chunk * SPW + slot;
    const int sub = LPS == 8 ? (lane & 3) + 4 * oct_half() : lane & (LPS - 1);
    const bool live = k < ((CCfg *)cfg)->K;  // (constant address space: a scalar load, not a register per lane)
    // instrumentation (mppi_set_wave_clock): the start stamp goes out now instead of occupying a register pair for the whole rollout
    if (wave_clk != nullptr && threadIdx.x == 0) {
#if defined(MPPI_HWID_DEBUG)
        wave_clk[2 * chunk] = hwdbg;
#else
        wave_clk[2 * chunk] = clk0;
#endif
    }
    // behind the SPW-wide sample rows: the wave-shared table of shape records and pair geometry blocks (scene_table_fill)
    CModel &M = *(CModel *)m;
    const int row = scene_row_floats<T>(M) + (NW == 2 ? scene_pair_floats<T>(M) : 0);
    unsigned *tab = reinterpret_cast<unsigned *>(lds + (size_t)SPW * row);
    scene_table_fill(M, tab, threadIdx.x, kWave * NW);
    LMem L{lds + slot, SPW, tab};
    L.cm = 11 * slot;
    if constexpr (NW == 1)
        if (M.n_light_pairs != 0) {   // records of the light bodies' pairs: the tail of the sample's rows (scene_row_floats)
            L.lp = lds + slot + (size_t)scene_light_base<T>(M) * SPW;
            L.lstride = SPW;
            light_region_reset(L);
        }
    // octet layout of the solve (mppi_scene_oct.hpp): the linear lanes read the bodies' inertia blocks from a copy without inertia
    // tensors; lane i stages body i
    constexpr bool kOctSolve = OSOLVE;
    __shared__ __attribute__((aligned(256))) uint4 s_lin_raw[(kOctSolve ? oct_lin_raw_bytes(T::NB) : 16) / 16];
    if constexpr (kOctSolve) {
        MPPI_LDS_AS DevBody *s_lin = oct_lin_place((MPPI_LDS_AS void *)s_lin_raw, &lm.b[0]);   // (bank-disjoint from the model's own blocks)
        if ((int)threadIdx.x < T::NB) {
            DevBody b = M.b[threadIdx.x];
            b.k1 = oct_lin_view(b.k1);
            s_lin[threadIdx.x] = b;
        }
        const MPPI_LDS_AS DevBody *mine = oct_half() ? (const MPPI_LDS_AS DevBody *)s_lin : &lm.b[0];
        L.oct_bodies = (unsigned)(unsigned long)mine;
    }
#if defined(MPPI_CHECK)
    L.limit = row;  // (floats of one sample's rows: any index beyond them belongs to the wave's table or to nobody)
#endif
    // start state in rollout coordinates (relative to the robot's start position, mppi_scene.hpp root_relative)
    __shared__ float s_root[13 * kMaxActors];
    root_origin(M, x0_root, L.ox, L.oy);
    for (int j = threadIdx.x; j < 13 * M.n_actors; j += kWave * NW) s_root[j] = root_relative_entry(x0_root, j, L.ox, L.oy);
    __syncthreads();
    if constexpr (NW == 2) {
        // (forward offsets only: an address formed as "end of the region minus a constant" cannot use the DS instructions'
        // unsigned immediate offset and costs a register per row - the compiler spilled those to scratch)
        L.set1 = scene_row_floats<T>(M);
        L.xch = L.set1 + SceneLayout<T>::NF * 27 + 3 * M.n_rb;
        L.park = L.xch + 2 + 6 * kFreeSlots;
        if (wave != 0) {
            // HELPER WAVEFRONT: its half of the shape poses and candidate pairs of every substep, out of and into LDS (see
            // kSplitOctPair); the barriers inside contact_forces pair with those of the first wavefront's calls
            if (live) {
                unsigned acc_dirty = ~0u, cf_dirty = ~0u;
                const int steps = cfg->H * M.substeps;
                for (int it = 0; it < steps; it++) {
                    contact_forces<T, kSplitOctPair>(*launder(&M), s_root, L, acc_dirty, cf_dirty, Split{sub, LPS, 1});
                    helper_free_bodies<T>(*launder(&M), L, Split{sub, LPS, 1});
                }
            }
            // the record tail's barriers are the workgroup's: the helper reaches them too and does none of the work
            quad_record<SPW>(*(CCfg *)cfg, INFINITY, false, du, chunk * SPW, partials, false, LPS == 8 ? slot : -1);
            fold_after_record<NW>(*(CCfg *)cfg, partials, fold_ctr, fold_out);
            return;
        }
    }
    float s = INFINITY;
    if (live) {
        s = rollout_scene<T, kSplit, DUMP>(M, lm, *(CCfg *)cfg, *(CCost *)cost, x0_dof, s_root, U, eps, prior, du, viz, k, L, Split{sub, LPS}, traj);
        if (sub == 0) S[(unsigned)k] = s;
    }
    quad_record<SPW>(*(CCfg *)cfg, s, live && sub == 0, du, chunk * SPW, partials + (size_t)chunk * (2 + cfg->H * cfg->nu), true, LPS == 8 ? slot : -1);
    fold_after_record<NW>(*(CCfg *)cfg, partials, fold_ctr, fold_out);
    MPPI_SEC(10);
    if (wave_clk != nullptr && lane == 0) {  // instrumentation (mppi_set_wave_clock): this wavefront's residency
        wave_clk[2 * chunk + 1] = wall_clock64();
#if defined(MPPI_SECTION_CLOCKS)
        for (int j = 0; j < kSections; j++) wave_clk[2 * (size_t)gridDim.x + (size_t)chunk * kSections + j] = section_counters()[j];
#endif
    }
#endif
}

// env state of contact scenes in HBM (sample-minor): base [13][K], free [kFreeSlots*13][K], cf [n_rb*3][K]
template <class T>
__global__ __launch_bounds__(kWave) void k_sim_step_scene(const DevModel *__restrict__ m, const DevCfg *__restrict__ cfg, int mode, int t,
                                                          const float *__restrict__ u_ext, const float *__restrict__ x0_root,
                                                          const float *__restrict__ U, const float *__restrict__ eps, const float *__restrict__ prior,
                                                          float *__restrict__ du, float *__restrict__ ctrl, float *__restrict__ q_, float *__restrict__ qd_,
                                                          float *__restrict__ base_, float *__restrict__ fr_, float *__restrict__ cf_) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NB = T::NB;
    const int K = cfg->K, nu = cfg->nu;
    const int k = blockIdx.x * kWave + threadIdx.x;
    if (k >= K) return;
    LMem L{lds + threadIdx.x, kWave};
    float light_rec[kLightFloats];   // (records of the light bodies' pairs: per lane, as in k_rollout_scene)
    L.lp = light_rec;
    L.lstride = 1;
    CModel &M = *(CModel *)m;
    if (M.n_light_pairs != 0) light_region_reset(L);
    SceneState<T> s;
    static_for<0, NB>([&](auto ic) {
        constexpr int i = ic;
        s.q[i] = q_[(size_t)i * K + k];
        s.qd[i] = qd_[(size_t)i * K + k];
    });
    static_for<0, T::NBASE>([&](auto rc) {   // (rows r * 13 + j: one block of 13 per base of the forest)
        constexpr int r = rc;
        for (int j = 0; j < 13; j++) s.template base_row<r>()[j] = base_[(size_t)(13 * r + j) * K + k];
    });
    for (int f = 0; f < kFreeSlots; f++)
        for (int j = 0; j < 13; j++) s.fr[f][j] = fr_[(size_t)(f * 13 + j) * K + k];
    float target[NB ? NB : 1], u[kMaxNu];
    const int g = cfg->k_offset + k;
    scene_randomise<T>(M, g, L);  // LDS does not persist across launches: redraw this sample's actor noise
    float cc = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxNu; c++) {
        float v = 0.f;
        if (c < nu) {
            if (mode == 0) v = u_ext[(size_t)k * nu + c];
            else if (mode == 1) v = u_ext[c];
            else {
                float Ut = U[t * nu + c];
                v = Ut + eps[(size_t)(t * nu + c) * K + k];
                if (cfg->sample_null_action && g == cfg->k_total - 1) v = 0.f;
                if (cfg->use_priors && prior != nullptr && g == cfg->k_total - 2) v = prior[t * nu + c];
                v = fminf(fmaxf(v, cfg->u_min.v[c]), cfg->u_max.v[c]);
                float d = v - Ut;
                du[(size_t)(t * nu + c) * K + k] = d;
                float term = Ut * d * cfg->inv_sigma.v[c];
                cc += cfg->lambda * (cfg->noise_abs_cost ? fabsf(term) : term);
            }
        }
        u[c] = v;
    }
    if (mode == 2) ctrl[k] += cc;
    cmd_map<T>(M, u, target);
    step_scene<T>(M, x0_root, s, target, L);
    static_for<0, NB>([&](auto ic) {
        constexpr int i = ic;
        q_[(size_t)i * K + k] = s.q[i];
        qd_[(size_t)i * K + k] = s.qd[i];
    });
    static_for<0, T::NBASE>([&](auto rc) {
        constexpr int r = rc;
        for (int j = 0; j < 13; j++) base_[(size_t)(13 * r + j) * K + k] = s.template base_row<r>()[j];
    });
    for (int f = 0; f < kFreeSlots; f++)
        for (int j = 0; j < 13; j++) fr_[(size_t)(f * 13 + j) * K + k] = s.fr[f][j];
    for (int j = 0; j < 3 * M.n_rb; j++) cf_[(size_t)j * K + k] = L[SceneLayout<T>::kCf + j];
}

// the same step with 4 lanes per env (generic Objective mode of a planner context: K/16 wavefronts, contact work dealt
// over the quad, quad-layout robot algebra for longer trees).  LDS does not persist across launches: the sample's noise
// draws and the static shapes' poses are rebuilt here.
template <class T>
__global__ __launch_bounds__(kWave) void k_sim_step_scene_quad(const DevModel *__restrict__ m, const DevCfg *__restrict__ cfg, int mode, int t,
                                                               const float *__restrict__ u_ext, const float *__restrict__ x0_root,
                                                               const float *__restrict__ U, const float *__restrict__ eps, const float *__restrict__ prior,
                                                               float *__restrict__ du, float *__restrict__ ctrl, float *__restrict__ q_, float *__restrict__ qd_,
                                                               float *__restrict__ base_, float *__restrict__ fr_, float *__restrict__ cf_,
                                                               float *__restrict__ fb_dof = nullptr, float *__restrict__ fb_root = nullptr) {
    // fb_dof / fb_root (K = 1 world of a closed loop): the stepped state goes straight into the PLANNER's next start state
    // (x0_dof [2n] interleaved, x0_root [A][13]) - what k_state_from_world, a device-to-device copy and k_root_from_world did
    // in three more launches behind this one
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NB = T::NB;
    constexpr int kModelBytes = (int)((offsetof(DevModel, sh) + 15) / 16 * 16);  // header, bodies, links, free bodies
    __shared__ __attribute__((aligned(64))) uint4 s_model[kModelBytes / 16];
    for (int i = threadIdx.x; i < kModelBytes / 16; i += kWave) s_model[i] = reinterpret_cast<const uint4 *>(m)[i];
    __syncthreads();
    LModel &lm = *(LModel *)s_model;
    const int K = cfg->K, nu = cfg->nu;
    const int k = blockIdx.x * 16 + (threadIdx.x >> 2);
    if (k >= K) return;
    const int lane4 = threadIdx.x & 3;
    const bool leader = lane4 == 0;
    const Split split{lane4, 4};
    LMem L{lds + (threadIdx.x >> 2), 16};
    L.cm = 11 * (int)(threadIdx.x >> 2);
    CModel &M = *(CModel *)m;
    if (M.n_light_pairs != 0) {   // records of the light bodies' pairs: the tail of the sample's rows (scene_row_floats)
        L.lp = lds + (threadIdx.x >> 2) + (size_t)scene_light_base<T>(M) * 16;
        L.lstride = 16;
        light_region_reset(L);
    }
    SceneState<T> s;
    static_for<0, NB>([&](auto ic) {
        constexpr int i = ic;
        s.q[i] = q_[(size_t)i * K + k];
        s.qd[i] = qd_[(size_t)i * K + k];
    });
    for (int j = 0; j < 13; j++) s.base[j] = base_[(size_t)j * K + k];
    for (int f = 0; f < kFreeSlots; f++)
        for (int j = 0; j < 13; j++) s.fr[f][j] = fr_[(size_t)(f * 13 + j) * K + k];
    float target[NB ? NB : 1], u[kMaxNu];
    const int g = cfg->k_offset + k;
    scene_randomise<T>(M, g, L);
    shape_cache_update<T>(M, x0_root, L, split, true);
    float cc = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxNu; c++) {
        float v = 0.f;
        if (c < nu) {
            if (mode == 0) v = u_ext[(size_t)k * nu + c];
            else if (mode == 1) v = u_ext[c];
            else {
                float Ut = U[t * nu + c];
                v = Ut + eps[(size_t)(t * nu + c) * K + k];
                if (cfg->sample_null_action && g == cfg->k_total - 1) v = 0.f;
                if (cfg->use_priors && prior != nullptr && g == cfg->k_total - 2) v = prior[t * nu + c];
                v = fminf(fmaxf(v, cfg->u_min.v[c]), cfg->u_max.v[c]);
                float d = v - Ut;
                if (leader) du[(size_t)(t * nu + c) * K + k] = d;
                float term = Ut * d * cfg->inv_sigma.v[c];
                cc += cfg->lambda * (cfg->noise_abs_cost ? fabsf(term) : term);
            }
        }
        u[c] = v;
    }
    if (mode == 2 && leader) ctrl[k] += cc;
    cmd_map<T>(M, u, target);
    step_scene_any<T, kSplitQuad>(M, lm, x0_root, s, target, L, split);
    if (leader) {
        static_for<0, NB>([&](auto ic) {
            constexpr int i = ic;
            q_[(size_t)i * K + k] = s.q[i];
            qd_[(size_t)i * K + k] = s.qd[i];
        });
        for (int j = 0; j < 13; j++) base_[(size_t)j * K + k] = s.base[j];
        for (int f = 0; f < kFreeSlots; f++)
            for (int j = 0; j < 13; j++) fr_[(size_t)(f * 13 + j) * K + k] = s.fr[f][j];
        for (int j = 0; j < 3 * M.n_rb; j++) cf_[(size_t)j * K + k] = L[SceneLayout<T>::kCf + j];
        if (fb_dof != nullptr && k == 0) {
            static_for<0, NB>([&](auto ic) {
                constexpr int i = ic;
                fb_dof[2 * i] = s.q[i];
                fb_dof[2 * i + 1] = s.qd[i];
            });
            for (int j = 0; j < 13 * M.n_actors; j++) fb_root[j] = x0_root[j];   // static actors: as the world holds them
            for (int j = 0; j < 13; j++) fb_root[13 * M.robot_actor + j] = s.base[j];
            for (int f = 0; f < kFreeSlots; f++)
                if (f < M.n_free)
                    for (int j = 0; j < 13; j++) fb_root[13 * M.fr[f].actor + j] = s.fr[f][j];
        }
    }
#endif
}

// K = 1 world (ABI 8, mppi_sim_materialise with the state mirror armed): env 0's dof / root rows, just written to the state tensors,
// also go into the context's mapped host block, followed by the sequence number behind a system-scope release - the host takes
// its torch.save payloads from there (mppi_mirror_wait), no device-to-host copy
constexpr int kIoMirrorRoot = 32;  // = kIoDofFloats (mppi_ctx hand-over block): the root rows follow the dof rows at this offset
__device__ __forceinline__ void mirror_env0(float *__restrict__ mirror, unsigned *__restrict__ mirror_seq, unsigned seq, int k, const float *__restrict__ dof,
                                            const float *__restrict__ root, int n_dof, int n_root) {
    if (mirror == nullptr || k != 0 || dof == nullptr || root == nullptr) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // this lane's own stores to dof / root are what it reads back
    for (int i = 0; i < n_dof; i++) mirror[i] = dof[i];
    for (int i = 0; i < n_root; i++) mirror[kIoMirrorRoot + i] = root[i];
    __threadfence_system();
    __hip_atomic_store(mirror_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <class T>
__global__ __launch_bounds__(kWave) void k_materialise_scene(const DevModel *__restrict__ m, int K, const float *__restrict__ x0_root,
                                                             const float *__restrict__ q_, const float *__restrict__ qd_, const float *__restrict__ base_,
                                                             const float *__restrict__ fr_, const float *__restrict__ cf_, float *__restrict__ dof,
                                                             float *__restrict__ root, float *__restrict__ rb, float *__restrict__ cf,
                                                             float *__restrict__ mirror = nullptr, unsigned *__restrict__ mirror_seq = nullptr, unsigned seq = 0) {
    constexpr int NB = T::NB;
    const int k = blockIdx.x * kWave + threadIdx.x;
    if (k >= K) return;
    CModel &M = *(CModel *)m;
    SceneState<T> s;
    static_for<0, NB>([&](auto ic) {
        constexpr int i = ic;
        s.q[i] = q_[(size_t)i * K + k];
        s.qd[i] = qd_[(size_t)i * K + k];
        if (dof != nullptr) {
            dof[(size_t)k * 2 * NB + 2 * i] = s.q[i];
            dof[(size_t)k * 2 * NB + 2 * i + 1] = s.qd[i];
        }
    });
    static_for<0, T::NBASE>([&](auto rc) {
        constexpr int r = rc;
        for (int j = 0; j < 13; j++) s.template base_row<r>()[j] = base_[(size_t)(13 * r + j) * K + k];
    });
    for (int f = 0; f < kFreeSlots; f++)
        for (int j = 0; j < 13; j++) s.fr[f][j] = fr_[(size_t)(f * 13 + j) * K + k];
    const int A = M.n_actors, B = M.n_rb;
    scene_materialise<T>(M, x0_root, s, nullptr, root != nullptr ? root + (size_t)k * 13 * A : nullptr,
                         rb != nullptr ? rb + (size_t)k * 13 * B : nullptr, nullptr);
    if (cf != nullptr)
        for (int j = 0; j < 3 * B; j++) cf[(size_t)k * 3 * B + j] = cf_[(size_t)j * K + k];
    mirror_env0(mirror, mirror_seq, seq, k, dof, root, 2 * NB, 13 * A);
}

// all envs <- x0 (root rows of the robot base and the free actors)
// (free_slots: the free-actor slots of the context's scene kernels - TopoEntry.free_slots -, NOT this unit's kFreeSlots: the
// kernels of a tree built on demand may carry four, mppi_hip.hip jit_topology)
__global__ void k_sim_reset_scene(const DevModel *__restrict__ m, int K, const float *__restrict__ x0_root, float *__restrict__ base_,
                                  float *__restrict__ fr_, float *__restrict__ cf_, int free_slots) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    for (int r = 0; r < m->n_bases; r++) {
        const int actor = r == 0 ? m->robot_actor : m->xbase_actor[r - 1];
        for (int j = 0; j < 13; j++) base_[(size_t)(13 * r + j) * K + k] = x0_root[13 * actor + j];
    }
    for (int f = 0; f < free_slots; f++)
        for (int j = 0; j < 13; j++) fr_[(size_t)(f * 13 + j) * K + k] = f < m->n_free ? x0_root[13 * m->fr[f].actor + j] : 0.f;
    for (int j = 0; j < 3 * m->n_rb; j++) cf_[(size_t)j * K + k] = 0.f;
}
// planner.x0_root rows of the robot base / free actors <- world env 0
__global__ void k_root_from_world(const DevModel *__restrict__ m, const float *__restrict__ wbase, const float *__restrict__ wfr, float *__restrict__ x0_root) {
    const int j = threadIdx.x;
    if (j < 13) {
        for (int r = 0; r < m->n_bases; r++) x0_root[13 * (r == 0 ? m->robot_actor : m->xbase_actor[r - 1]) + j] = wbase[13 * r + j];
        for (int f = 0; f < m->n_free; f++) x0_root[13 * m->fr[f].actor + j] = wfr[f * 13 + j];
    }
}

// ---- halton-spline sampler -------------------------------------------------------------------
__constant__ int c_primes[MPPI_MAX_KNOTS * MPPI_MAX_NU] = {
    2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 101, 103, 107, 109, 113,
    127, 131, 137, 139, 149, 151, 157, 163, 167, 173, 179, 181, 191, 193, 197, 199, 211, 223, 227, 229, 233, 239, 241, 251,
    257, 263, 269, 271, 277, 281, 283, 293, 307, 311, 313, 317, 331, 337, 347, 349, 353, 359, 367, 373, 379, 383, 389, 397,
    401, 409, 419, 421, 431, 433, 439, 443, 449, 457, 461, 463, 467, 479, 487, 491, 499, 503, 509, 521, 523, 541, 547, 557,
    563, 569, 571, 577, 587, 593, 599, 601, 607, 613, 617, 619, 631, 641, 643, 647, 653, 659, 661, 673, 677, 683, 691, 701,
    709, 719, 727, 733, 739, 743, 751, 757, 761, 769, 773, 787, 797, 809, 811, 821, 823, 827, 829, 839, 853, 857, 859, 863,
    877, 881, 883, 887, 907, 911, 919, 929, 937, 941, 947, 953, 967, 971, 977, 983, 991, 997, 1009, 1013, 1019, 1021, 1031,
    1033, 1039, 1049, 1051, 1061, 1063, 1069, 1087, 1091, 1093, 1097, 1103, 1109, 1117, 1123, 1129, 1151, 1153, 1163};

// linearly digit-scrambled radical inverse in double (digit -> digit*mult mod p, mult = round(0.618 p))
__device__ double halton_scrambled(uint32_t n, int dim) {
    const uint32_t p = (uint32_t)c_primes[dim];
    uint32_t mult = (uint32_t)(0.6180339887498949 * (double)p + 0.5);
    if (mult < 1u) mult = 1u;
    double f = 1.0 / (double)p, r = 0.0;
    const double invp = f;
    while (n > 0u) {
        uint32_t dgt = n % p;
        r += f * (double)((dgt * mult) % p);
        n /= p;
        f *= invp;
    }
    return r;
}

// eps[(t*nu+c)*K + k] = sigma_c * sum_i B[t][i] * Phi^-1(halton(g+1+base, i*nu+c))
__global__ __launch_bounds__(kWave) void k_sample(const DevCfg *__restrict__ cfg, const double *__restrict__ basis, const double *__restrict__ sigma,
                                                  int n_knots, uint32_t index_base, float *__restrict__ eps) {
    const int K = cfg->K, H = cfg->H, nu = cfg->nu;
    const int k = blockIdx.x * kWave + threadIdx.x;
    if (k >= K) return;
    const uint32_t n = (uint32_t)(cfg->k_offset + k) + 1u + index_base;
    for (int c = 0; c < nu; c++) {
        double z[MPPI_MAX_KNOTS];
        for (int i = 0; i < n_knots; i++) z[i] = normcdfinv(halton_scrambled(n, i * nu + c));
        for (int t = 0; t < H; t++) {
            double s = 0.0;
            for (int i = 0; i < n_knots; i++) s += basis[t * n_knots + i] * z[i];
            eps[(size_t)(t * nu + c) * K + k] = (float)(sigma[c] * s);
        }
    }
}

// ---- counter-based Gaussian sampler (MPPI_SAMPLE_NORMAL) ---------------------------------------
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): ten rounds of two 32x32->64
// multiplies and a key bump.  Known-answer vectors of the Random123 distribution are checked in tests/.
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *out) {
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
constexpr uint32_t kPhiloxKey1 = 0x4D505049u;  // "MPPI"

// One thread per (sample k, control dimension c); the thread index is k-minor so every store is coalesced.
// z[i] (knot i of control c of GLOBAL sample g): words (2p, 2p+1) of philox(counter = (g, c, i/4, iteration),
// key = (seed, "MPPI")) -> u = (x + 0.5) 2^-32 -> Box-Muller pair (r cos, r sin), r = sqrt(-2 ln u_a), angle 2 pi u_b.
// eps[(t*nu+c)*K + k] = mu_c + sigma_c * sum_i B[t][i] z[i]   (n_knots == H: no spline, eps_t = mu + sigma z_t)
__global__ __launch_bounds__(kWave) void k_sample_normal(const DevCfg *__restrict__ cfg, const double *__restrict__ basis,
                                                         const double *__restrict__ sigma_mu, int n_knots, uint32_t seed, uint32_t iteration,
                                                         float *__restrict__ eps) {
    const int K = cfg->K, H = cfg->H, nu = cfg->nu;
    const int idx = blockIdx.x * kWave + threadIdx.x;
    if (idx >= K * nu) return;
    const int c = idx / K, k = idx - c * K;
    const uint32_t g = (uint32_t)(cfg->k_offset + k);
    const double sg = sigma_mu[c], mu = sigma_mu[MPPI_MAX_NU + c];
    const bool direct = n_knots == H;
    double z[MPPI_MAX_KNOTS];
    for (int b = 0; 4 * b < n_knots; b++) {
        uint32_t x[4];
        philox4x32_10(g, (uint32_t)c, (uint32_t)b, iteration, seed, kPhiloxKey1, x);
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const double ua = ((double)x[2 * p] + 0.5) * (1.0 / 4294967296.0), ub = ((double)x[2 * p + 1] + 0.5) * (1.0 / 4294967296.0);
            const double r = sqrt(-2.0 * log(ua));
            double sn, cs;
            sincos(6.283185307179586 * ub, &sn, &cs);
            const double zz[2] = {r * cs, r * sn};
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int i = 4 * b + 2 * p + e;
                if (i < n_knots) {
                    if (direct) eps[(size_t)(i * nu + c) * K + k] = (float)(mu + sg * zz[e]);
                    else z[i] = zz[e];
                }
            }
        }
    }
    if (!direct)
        for (int t = 0; t < H; t++) {
            double s = 0.0;
            for (int i = 0; i < n_knots; i++) s += basis[t * n_knots + i] * z[i];
            eps[(size_t)(t * nu + c) * K + k] = (float)(mu + sg * s);
        }
}

// ---- batched simulator (generic Objective mode, K=1 world) -------------------------------------
__global__ void k_sim_reset(int K, int n, const float *__restrict__ x0_dof, float *__restrict__ q, float *__restrict__ qd,
                            float *__restrict__ S, float *__restrict__ ctrl) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    for (int i = 0; i < n; i++) {
        q[(size_t)i * K + k] = x0_dof[2 * i];
        qd[(size_t)i * K + k] = x0_dof[2 * i + 1];
    }
    S[k] = 0.f;
    ctrl[k] = 0.f;
}

// mode 0: u_ext is [K][nu] (reference layout), mode 1: u_ext is one shared [nu], mode 2: horizon step t
template <class T>
__global__ __launch_bounds__(kWave) void k_sim_step(const DevModel *__restrict__ m, const DevCfg *__restrict__ cfg, int mode, int t,
                                                    const float *__restrict__ u_ext, const float *__restrict__ x0_root,
                                                    const float *__restrict__ U, const float *__restrict__ eps, const float *__restrict__ prior,
                                                    float *__restrict__ du, float *__restrict__ ctrl, float *__restrict__ q_, float *__restrict__ qd_) {
    constexpr int NB = T::NB;
    const int K = cfg->K, nu = cfg->nu;
    const int k = blockIdx.x * kWave + threadIdx.x;
    if (k >= K) return;
    float q[NB], qd[NB], target[NB], u[kMaxNu];
    static_for<0, NB>([&](auto ic) {
        constexpr int i = ic;
        q[i] = q_[(size_t)i * K + k];
        qd[i] = qd_[(size_t)i * K + k];
    });
    const int g = cfg->k_offset + k;
    float cc = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxNu; c++) {
        float v = 0.f;
        if (c < nu) {
            if (mode == 0) v = u_ext[(size_t)k * nu + c];
            else if (mode == 1) v = u_ext[c];
            else {
                float Ut = U[t * nu + c];
                v = Ut + eps[(size_t)(t * nu + c) * K + k];
                if (cfg->sample_null_action && g == cfg->k_total - 1) v = 0.f;
                if (cfg->use_priors && prior != nullptr && g == cfg->k_total - 2) v = prior[t * nu + c];
                v = fminf(fmaxf(v, cfg->u_min.v[c]), cfg->u_max.v[c]);
                float d = v - Ut;
                du[(size_t)(t * nu + c) * K + k] = d;
                float term = Ut * d * cfg->inv_sigma.v[c];
                cc += cfg->lambda * (cfg->noise_abs_cost ? fabsf(term) : term);
            }
        }
        u[c] = v;
    }
    if (mode == 2) ctrl[k] += cc;
    cmd_map<T>(*(CModel *)m, u, target);
    step<T>(*(CModel *)m, x0_root, q, qd, target);
    static_for<0, NB>([&](auto ic) {
        constexpr int i = ic;
        q_[(size_t)i * K + k] = q[i];
        qd_[(size_t)i * K + k] = qd[i];
    });
}

// the same step with one sample per 4-lane quad (generic Objective mode of the planner: K/16 wavefronts instead of K/64)
template <class T>
__global__ __launch_bounds__(kWave) void k_sim_step_quad(const DevModel *__restrict__ m, const DevCfg *__restrict__ cfg, int mode, int t,
                                                         const float *__restrict__ u_ext, const float *__restrict__ x0_root,
                                                         const float *__restrict__ U, const float *__restrict__ eps, const float *__restrict__ prior,
                                                         float *__restrict__ du, float *__restrict__ ctrl, float *__restrict__ q_, float *__restrict__ qd_) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NB = T::NB;
    constexpr int kModelBytes = (int)((offsetof(DevModel, sh) + 15) / 16 * 16);  // header, bodies, links, free bodies
    __shared__ __attribute__((aligned(64))) uint4 s_model[kModelBytes / 16];
    for (int i = threadIdx.x; i < kModelBytes / 16; i += kWave) s_model[i] = reinterpret_cast<const uint4 *>(m)[i];
    __syncthreads();
    LModel &lm = *(LModel *)s_model;
    const int K = cfg->K, nu = cfg->nu;
    const int k = blockIdx.x * 16 + (threadIdx.x >> 2);
    if (k >= K) return;
    const bool leader = (threadIdx.x & 3) == 0;
    QF q[NB], qd[NB], target[NB];
    static_for<0, NB>([&](auto ic) {
        constexpr int i = ic;
        q[i] = q_[(size_t)i * K + k];
        qd[i] = qd_[(size_t)i * K + k];
    });
    const int g = cfg->k_offset + k;
    float u[kMaxNu], cc = 0.f;
#pragma unroll
    for (int c = 0; c < kMaxNu; c++) {
        float v = 0.f;
        if (c < nu) {
            if (mode == 0) v = u_ext[(size_t)k * nu + c];
            else if (mode == 1) v = u_ext[c];
            else {
                float Ut = U[t * nu + c];
                v = Ut + eps[(size_t)(t * nu + c) * K + k];
                if (cfg->sample_null_action && g == cfg->k_total - 1) v = 0.f;
                if (cfg->use_priors && prior != nullptr && g == cfg->k_total - 2) v = prior[t * nu + c];
                v = fminf(fmaxf(v, cfg->u_min.v[c]), cfg->u_max.v[c]);
                float d = v - Ut;
                if (leader) du[(size_t)(t * nu + c) * K + k] = d;
                float term = Ut * d * cfg->inv_sigma.v[c];
                cc += cfg->lambda * (cfg->noise_abs_cost ? fabsf(term) : term);
            }
        }
        u[c] = v;
    }
    if (mode == 2 && leader) ctrl[k] += cc;
    static_for<0, NB>([&](auto ic) {
        constexpr int i = ic;
        const CmdBlock b = load_block<CmdBlock>(lm.b[i].cmd);
        float tg = 0.f;
#pragma unroll
        for (int c = 0; c < kMaxNu; c++) tg += b.v[c] * u[c];
        target[i] = tg;
    });
    QPose<T> P;
    quad_base<T>(lm, x0_root, P);
    quad_fk<T>(lm, q, P);
    quad_step<T>(lm, P, q, qd, target);
    if (leader)
        static_for<0, NB>([&](auto ic) {
            constexpr int i = ic;
            q_[(size_t)i * K + k] = q[i];
            qd_[(size_t)i * K + k] = qd[i];
        });
#endif
}

// sample-minor sim state -> reference-layout tensors (isaacgym_wrapper.py:186-199)
template <class T>
__global__ __launch_bounds__(kWave) void k_materialise(const DevModel *__restrict__ m, int K, const float *__restrict__ x0_root,
                                                       const float *__restrict__ q_, const float *__restrict__ qd_, float *__restrict__ dof,
                                                       float *__restrict__ root, float *__restrict__ rb, float *__restrict__ cf,
                                                       float *__restrict__ mirror = nullptr, unsigned *__restrict__ mirror_seq = nullptr, unsigned seq = 0) {
    constexpr int NB = T::NB;
    const int k = blockIdx.x * kWave + threadIdx.x;
    if (k >= K) return;
    float q[NB ? NB : 1], qd[NB ? NB : 1];
    static_for<0, NB>([&](auto ic) {
        constexpr int i = ic;
        q[i] = q_[(size_t)i * K + k];
        qd[i] = qd_[(size_t)i * K + k];
        if (dof != nullptr) {
            dof[(size_t)k * 2 * NB + 2 * i] = q[i];
            dof[(size_t)k * 2 * NB + 2 * i + 1] = qd[i];
        }
    });
    const int A = m->n_actors, B = m->n_rb;
    if (root != nullptr)
        for (int j = 0; j < 13 * A; j++) root[(size_t)k * 13 * A + j] = x0_root[j];
    if (rb != nullptr) rigid_body_state<T>(*(CModel *)m, x0_root, q, qd, rb + (size_t)k * 13 * B, cf != nullptr ? cf + (size_t)k * 3 * B : nullptr);
    else if (cf != nullptr)
        for (int j = 0; j < 3 * B; j++) cf[(size_t)k * 3 * B + j] = 0.f;
    mirror_env0(mirror, mirror_seq, seq, k, dof, root, 2 * NB, 13 * A);
}

// one robot link of N envs as dense rows [N][13] (generic Objective mode: `sim.get_actor_link_by_name(..)` over the horizon view needs
// the rows of the ONE link it names, not the [N][n_rb][13] tensor - 68 us of kinematics and scattered stores for the panda's 11
// links against ~15 for one).  Each wavefront's 64 rows x 13 floats are 832 consecutive floats: staged in LDS, stored coalesced.
template <class T>
__global__ __launch_bounds__(kWave) void k_materialise_link(const DevModel *__restrict__ m, int N, const float *__restrict__ x0_root, const float *__restrict__ q_,
                                                            const float *__restrict__ qd_, int link, float *__restrict__ out) {
    constexpr int NB = T::NB;
    __shared__ float s_row[kWave * 13];
    const int k = blockIdx.x * kWave + threadIdx.x;
    if (k < N) {
        float q[NB ? NB : 1], qd[NB ? NB : 1], o[13];
        static_for<0, NB>([&](auto ic) {
            constexpr int i = ic;
            q[i] = q_[(size_t)i * N + k];
            qd[i] = qd_[(size_t)i * N + k];
        });
        rigid_body_link<T>(*(CModel *)m, x0_root, q, qd, link, o);
        for (int j = 0; j < 13; j++) s_row[threadIdx.x * 13 + j] = o[j];
    }
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * kWave * 13, end = (size_t)N * 13;
    for (int j = 0; j < 13; j++) {
        const size_t at = base + (size_t)j * kWave + threadIdx.x;
        if (at < end) out[at] = s_row[j * kWave + threadIdx.x];
    }
}

// Parity / debug entry (mppi_eval_cost): the cost program of the context evaluated by the interpreter the rollout kernels run
// (program_cost_with, mppi_device.hpp) on CALLER-GIVEN simulator answers - reference-layout rows of n envs - instead of the
// kernel's own kinematics: link poses from the rigid-body rows (position, quaternion xyzw), actor rows from the root rows,
// contact forces as given.  This is how the reference's golden Objective inputs (tests/golden/objective_costs.json: what the
// gym getters returned, reference examples/*/planner.py compute_cost) reach the HIP cost path directly.
struct GivenEnv {
    const float *root, *cfr;
    MPPI_HD V3 vec(int actor, int off) const { return loadv(root + 13 * actor + off); }
    MPPI_HD void quat(int actor, float *qq) const {
        for (int j = 0; j < 4; j++) qq[j] = root[13 * actor + 3 + j];
    }
    MPPI_HD float cf(int rb, int j) const { return cfr[3 * rb + j]; }
    MPPI_HD V3 constant_point(float x, float y, float z) const { return V3{x, y, z}; }
};
template <class T>
__global__ __launch_bounds__(kWave) void k_eval_cost(const DevModel *__restrict__ m_, const DevCost *__restrict__ cost_, int n, const float *__restrict__ dof,
                                                     const float *__restrict__ root, const float *__restrict__ rb, const float *__restrict__ cf,
                                                     float *__restrict__ out) {
    constexpr int NB = T::NB;
    CModel &m = *(CModel *)m_;
    CCost &c = *(CCost *)cost_;
    const int k = blockIdx.x * kWave + threadIdx.x;
    if (k >= n) return;
    float q[NB ? NB : 1], qd[NB ? NB : 1];
    static_for<0, NB>([&](auto ic) {
        constexpr int i = ic;
        q[i] = dof[(size_t)k * 2 * NB + 2 * i];
        qd[i] = dof[(size_t)k * 2 * NB + 2 * i + 1];
    });
    const float *rows = rb + (size_t)k * 13 * m.n_rb + 13 * m.robot_first_rb;
    const GivenEnv env{root + (size_t)k * 13 * m.n_actors, cf + (size_t)k * 3 * m.n_rb};
    out[k] = program_cost_with<T>(c, q, qd, [&](int l, M3 &R, V3 &p) MPPI_LAMBDA {
        p = loadv(rows + 13 * l);
        R = quat_to_R(rows + 13 * l + 3);
    }, env);
}

__global__ void k_accumulate_cost(int K, float disc, const float *__restrict__ c, float *__restrict__ S) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) S[k] += disc * c[k];
}
__global__ void k_sim_finish(int K, const float *__restrict__ ctrl, float *__restrict__ S) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) S[k] += ctrl[k];
}
// planner.x0_dof <- world env 0 (K_world = 1: sample-minor == plain arrays)
__global__ void k_state_from_world(int n, const float *__restrict__ wq, const float *__restrict__ wqd, float *__restrict__ x0_dof) {
    const int i = threadIdx.x;
    if (i < n) {
        x0_dof[2 * i] = wq[i];
        x0_dof[2 * i + 1] = wqd[i];
    }
}

}  // namespace

// host hand-over block of a context (mapped pinned memory, ABI 8): [0] sequence number of the state mirror, [kIoCmd ...] a
// ring of kIoCmdSlots commands of 16 floats (mppi_sim_step_host: the step kernel reads its command through the mapped pointer),
// [kIoDof ...] the mirrored dof state (2n <= kIoDofFloats) followed by the root state (13 A) of a K = 1 world
constexpr int kIoCmd = 16, kIoCmdSlots = 64, kIoDof = kIoCmd + 16 * kIoCmdSlots, kIoDofFloats = 32;
static_assert(kIoDofFloats == kIoMirrorRoot, "mirror layout");
constexpr int kIoFloats = kIoDof + kIoDofFloats + 13 * MPPI_MAX_ACTORS + 12;

// ------------------------------------------------------------------------------ context
struct mppi_ctx {
    int device = 0;
    float *h_io = nullptr, *d_io = nullptr;  // the hand-over block and its device address
    int free_slots = 2;  // free-actor slots of the scene kernels this context launches (TopoEntry.free_slots): sizes d_fr and the trajectory rows
    unsigned io_cmd_next = 0, io_mirror_seq = 0;
    bool mirror_armed = false;  // the next materialise launch also mirrors env 0 into h_io (mppi_sim_materialise_mirror)
    hipStream_t stream = nullptr;
    mppi_model_t model;
    mppi_config_t cfg;
    DevModel hm;
    DevCfg hc;
    DevCost hk;
    int n = 0, A = 0, B = 0, K = 0, H = 0, nu = 0, HN = 0, RF = 0, n_waves = 0;
    int n_quads = 0;      // wavefronts of the quad- / octet-parallel rollout (16 / 8 samples each)
    int lanes_per_sample = 1;  // 1 (lane kernels), 4 (quad kernels), 8 (contact scenes: octets)
    float *d_traj = nullptr;          // per-step env states of one rollout set (mppi_rollout_trajectory), allocated on first use
    DevCost *d_cost_none = nullptr;    // a zero cost for those rollouts
    void (*launch_rollout_traj)(mppi_ctx *) = nullptr;
    void (*launch_materialise_traj)(mppi_ctx *, float *, float *, float *, float *) = nullptr;
    void (*launch_materialise_traj_link)(mppi_ctx *, int, float *) = nullptr;  // one robot link of all H*K env-steps (contact-free scenes)
    bool helper_wave = false;  // octet rollout kernel with a second wavefront per sample group for half of the contact pairs
    int n_partials = 0;   // records currently held by d_partials
    bool quad = false;
    DevModel *d_model = nullptr;
    DevCfg *d_cfg = nullptr;
    DevCost *d_cost = nullptr;
    float *d_x0_dof = nullptr, *d_x0_root = nullptr, *d_U = nullptr, *d_eps = nullptr, *d_du = nullptr, *d_S = nullptr;
    float *d_prior = nullptr, *d_viz = nullptr, *d_partials = nullptr, *d_record = nullptr, *d_action = nullptr, *d_beta_eta = nullptr;
    float *d_q = nullptr, *d_qd = nullptr, *d_ctrl = nullptr;
    float *d_base = nullptr, *d_fr = nullptr, *d_cf = nullptr;  // contact scenes: env root rows and contact forces
    float *d_filter = nullptr;  // filter_u operator [H][H]
    float *h_action = nullptr;  // pinned, host-mapped mirror of d_action; word 16 = sequence number of the last update
    unsigned *d_seq = nullptr;
    unsigned seq_expected = 0;  // updates launched so far
    bool use_filter = false;
    bool scene = false;
    size_t lds_bytes = 0, lds_bytes_quad = 0;  // dynamic LDS of the lane-per-sample / quad-per-sample scene kernels
    size_t lds_bytes_table = 0;                // ... plus the wave-shared shape / pair table of the shared-lane rollout kernels
    size_t lds_bytes_static = 0;               // ... plus the kernels' static __shared__ (hipFuncGetAttributes)
    double *d_basis = nullptr, *d_sigma = nullptr;
    const float *eps_in = nullptr;  // d_eps or an external noise buffer
    bool has_prior = false, has_cost = false, profiling = false;
    bool partials_valid = false;  // d_partials holds the records of the current S (written by the fused rollout tail)
    // fold of the wave records inside the quad rollout kernels (fold_group): per-group counters, the folded records
    // (own buffer d_fold, or the caller's mppi_set_record_out buffer - e.g. this rank's rows of the all-gather tensor)
    bool fold = false;
    unsigned *d_fold_ctr = nullptr;
    float *d_fold = nullptr, *fold_out = nullptr;
    const float *recs_cur = nullptr;  // records the next combine reads when the caller passes none
    unsigned long long *d_wave_clk = nullptr;  // instrumentation: [wavefront][start, end] of the last quad rollout
    bool wave_clk_on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[3];
    size_t ev_used[3] = {0, 0, 0}, ev_seen[3] = {0, 0, 0};
    int profile_period = 1;  // hipEvent brackets on every n-th launch
    void (*launch_rollout)(mppi_ctx *) = nullptr;
    void (*launch_rollout_lane)(mppi_ctx *) = nullptr;  // one-lane kernel of a contact-free scene (cost programs)
    bool prog_lane = false;
    void (*launch_sim_step)(mppi_ctx *, int, int, const float *) = nullptr;
    void (*launch_materialise)(mppi_ctx *, float *, float *, float *, float *) = nullptr;
    void (*launch_combine_world)(mppi_ctx *, const float *, int, mppi_ctx *) = nullptr;  // fused closed-loop tail (quad scenes)
    void (*launch_eval_cost)(mppi_ctx *, int, const float *, const float *, const float *, const float *, float *) = nullptr;
    std::string topo;
    // closed loop: where the K = 1 world's step kernel writes the planner's next start state (mppi_update_step_world; null otherwise)
    float *fb_dof = nullptr, *fb_root = nullptr;
    bool step_feeds_back = false;  // the context's step kernel honours fb_dof / fb_root (the quad step kernel of contact scenes)
    // direct exchange of the shard records (mppi_mailbox_*): this rank's inbox, the peers' inboxes, the gathered records
    void *d_inbox = nullptr;
    void **d_peers = nullptr;
    std::vector<void *> h_peers, ipc_opened;
    int mb_rank = -1, mb_n = 0, mb_nrec = 0;
    unsigned *d_mb_seq = nullptr, *d_mb_status = nullptr;
    float *d_gathered = nullptr, *d_own_rec = nullptr;
    size_t inbox_bytes = 0;
    bool peers_dirty = false;
    bool inbox_fine = false;  // the inbox is fine-grained device memory (what peers of another GPU / process need)
};

// one row of the launch table: the kinematic tree and its kernel launchers
namespace mppi {
struct TopoEntry {
    int nb;
    int parents[MPPI_MAX_BODIES];
    size_t scene_lds_floats;  // per-lane LDS floats of the contact-scene kernels, excluding 3 * n_rb
    int free_slots;           // free-actor slots of the contact-scene kernels (kFreeSlots of the unit that filled them; 0: none filled)
    void (*rollout)(mppi_ctx *);
    void (*rollout_quad)(mppi_ctx *);
    void (*rollout_oct)(mppi_ctx *);        // contact-free scenes, octet layout of the solve (8 lanes per sample)
    void (*rollout_oct_traj)(mppi_ctx *);
    void (*rollout_scene)(mppi_ctx *);
    void (*rollout_scene_quad)(mppi_ctx *);
    void (*rollout_scene_oct)(mppi_ctx *);  // 8 lanes per sample
    void (*rollout_scene_pair)(mppi_ctx *); // 8 lanes per sample + a helper wavefront (null: trees of more than 4 bodies)
    void (*rollout_traj)(mppi_ctx *);        // fused rollouts with the per-step states dumped (generic Objective mode)
    void (*rollout_scene_traj)(mppi_ctx *);
    void (*materialise_traj)(mppi_ctx *, float *, float *, float *, float *);
    void (*materialise_traj_link)(mppi_ctx *, int, float *);
    void (*materialise_scene_traj)(mppi_ctx *, float *, float *, float *, float *);
    void (*sim_step)(mppi_ctx *, int, int, const float *);
    void (*sim_step_quad)(mppi_ctx *, int, int, const float *);
    void (*sim_step_scene)(mppi_ctx *, int, int, const float *);
    void (*sim_step_scene_quad)(mppi_ctx *, int, int, const float *);
    void (*materialise)(mppi_ctx *, float *, float *, float *, float *);
    void (*materialise_scene)(mppi_ctx *, float *, float *, float *, float *);
    void (*combine_world)(mppi_ctx *, const float *, int, mppi_ctx *);
    void (*eval_cost)(mppi_ctx *, int, const float *, const float *, const float *, const float *, float *);  // mppi_eval_cost
    hipError_t (*raise_lds)(size_t, size_t);
    size_t (*static_lds)();  // largest static __shared__ footprint among the contact-scene kernels (counts against the 160 KiB too)
};
}  // namespace mppi

namespace {

template <class T>
void launch_rollout_scene_t(mppi_ctx *c) {
    hipLaunchKernelGGL(k_rollout_scene<T>, dim3(c->n_waves), dim3(kWave), c->lds_bytes, c->stream, c->d_model, c->d_cfg, c->d_cost, c->d_x0_dof,
                       c->d_x0_root, c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_S, c->cfg.want_rollouts ? c->d_viz : nullptr,
                       c->d_partials);
}
// the octet kernel of a fixed-base tree of more than four bodies runs the solve in the octet layout as well (MPPI_SCENE_SOLVE=quad:
// the quad-layout solve in both quads, for A/B measurements)
template <class T>
constexpr bool kHasOctSolve = (T::NB > 4);
inline bool scene_oct_solve(const mppi_ctx *c) {
    const char *e = std::getenv("MPPI_SCENE_SOLVE");
    return c->hm.floating == 0 && !(e && std::string(e) == "quad");
}
template <class T, int LPS>
void launch_rollout_scene_quad_t(mppi_ctx *c) {
    if constexpr (LPS == 8 && kHasOctSolve<T>) {
        if (scene_oct_solve(c)) {
            hipLaunchKernelGGL((k_rollout_scene_quad<T, 8, 1, false, true>), dim3(c->n_quads), dim3(kWave), c->lds_bytes_quad * (kWave / LPS) / 16 + c->lds_bytes_table, c->stream, c->d_model, c->d_cfg, c->d_cost, c->d_x0_dof,
                               c->d_x0_root, c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_S, c->cfg.want_rollouts ? c->d_viz : nullptr,
                               c->d_partials, c->fold ? c->d_fold_ctr : nullptr, c->fold_out, c->wave_clk_on ? c->d_wave_clk : nullptr);
            return;
        }
    }
    hipLaunchKernelGGL((k_rollout_scene_quad<T, LPS>), dim3(c->n_quads), dim3(kWave), c->lds_bytes_quad * (kWave / LPS) / 16 + c->lds_bytes_table, c->stream, c->d_model, c->d_cfg, c->d_cost, c->d_x0_dof,
                       c->d_x0_root, c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_S, c->cfg.want_rollouts ? c->d_viz : nullptr,
                       c->d_partials, c->fold ? c->d_fold_ctr : nullptr, c->fold_out, c->wave_clk_on ? c->d_wave_clk : nullptr);
}
// octet layout with a helper wavefront per sample group (short trees: kSplitOctPair)
template <class T>
size_t pair_lds_bytes(const mppi_ctx *c) {
    const size_t row = c->lds_bytes_quad / 16 + sizeof(float) * ((size_t)SceneLayout<T>::NF * 27 + 3 * (size_t)c->hm.n_rb + 2 + 6 * kFreeSlots + scene_park_floats<T>());
    return row * (kWave / 8) + c->lds_bytes_table;
}
template <class T>
void launch_rollout_scene_pair_t(mppi_ctx *c) {
    if constexpr (T::NB <= 4) {
        hipLaunchKernelGGL((k_rollout_scene_quad<T, 8, 2>), dim3(c->n_quads), dim3(2 * kWave), pair_lds_bytes<T>(c), c->stream, c->d_model, c->d_cfg, c->d_cost, c->d_x0_dof,
                           c->d_x0_root, c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_S, c->cfg.want_rollouts ? c->d_viz : nullptr,
                           c->d_partials, c->fold ? c->d_fold_ctr : nullptr, c->fold_out, c->wave_clk_on ? c->d_wave_clk : nullptr);
    }
}
// generic Objective mode, whole horizon at once: the fused rollout with the per-step states dumped (cost NONE), then the
// reference-layout tensors of all H*K env-steps from ONE materialise launch
template <class T>
void launch_rollout_scene_traj_t(mppi_ctx *c) {
    if constexpr (T::NB <= 4) {  // short trees: the kernel with the helper wavefront (pair_lds_bytes is defined below)
        if (c->helper_wave) {
            const size_t row = c->lds_bytes_quad / 16 + sizeof(float) * ((size_t)SceneLayout<T>::NF * 27 + 3 * (size_t)c->hm.n_rb + 2 + 6 * kFreeSlots + scene_park_floats<T>());
            hipLaunchKernelGGL((k_rollout_scene_quad<T, 8, 2, true>), dim3(c->n_quads), dim3(2 * kWave), row * (kWave / 8) + c->lds_bytes_table, c->stream, c->d_model, c->d_cfg, c->d_cost_none,
                               c->d_x0_dof, c->d_x0_root, c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_S, (float *)nullptr,
                               c->d_partials, (unsigned *)nullptr, c->fold_out, (unsigned long long *)nullptr, c->d_traj);
            return;
        }
    }
    if constexpr (kHasOctSolve<T>) {
        if (scene_oct_solve(c)) {
            hipLaunchKernelGGL((k_rollout_scene_quad<T, 8, 1, true, true>), dim3(c->n_quads), dim3(kWave), c->lds_bytes_quad * (kWave / 8) / 16 + c->lds_bytes_table, c->stream, c->d_model, c->d_cfg, c->d_cost_none, c->d_x0_dof,
                               c->d_x0_root, c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_S, (float *)nullptr,
                               c->d_partials, (unsigned *)nullptr, c->fold_out, (unsigned long long *)nullptr, c->d_traj);
            return;
        }
    }
    hipLaunchKernelGGL((k_rollout_scene_quad<T, 8, 1, true>), dim3(c->n_quads), dim3(kWave), c->lds_bytes_quad * (kWave / 8) / 16 + c->lds_bytes_table, c->stream, c->d_model, c->d_cfg, c->d_cost_none, c->d_x0_dof,
                       c->d_x0_root, c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_S, (float *)nullptr,
                       c->d_partials, (unsigned *)nullptr, c->fold_out, (unsigned long long *)nullptr, c->d_traj);
}
template <class T>
void launch_materialise_scene_traj_t(mppi_ctx *c, float *dof, float *root, float *rb, float *cf) {
    const size_t HK = (size_t)c->H * c->K;
    float *q = c->d_traj, *qd = q + (size_t)T::NB * HK, *base = qd + (size_t)T::NB * HK, *fr = base + 13 * HK, *cfr = fr + (size_t)13 * kFreeSlots * HK;
    hipLaunchKernelGGL(k_materialise_scene<T>, dim3((unsigned)((HK + kWave - 1) / kWave)), dim3(kWave), 0, c->stream, c->d_model, (int)HK, c->d_x0_root, q, qd, base,
                       fr, cfr, dof, root, rb, cf);
}
template <class T>
void launch_sim_step_scene_t(mppi_ctx *c, int mode, int t, const float *u_ext) {
    hipLaunchKernelGGL(k_sim_step_scene<T>, dim3(c->n_waves), dim3(kWave), c->lds_bytes, c->stream, c->d_model, c->d_cfg, mode, t, u_ext, c->d_x0_root,
                       c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_ctrl, c->d_q, c->d_qd, c->d_base, c->d_fr, c->d_cf);
}
template <class T>
void launch_sim_step_scene_quad_t(mppi_ctx *c, int mode, int t, const float *u_ext) {
    hipLaunchKernelGGL(k_sim_step_scene_quad<T>, dim3((c->K + 15) / 16), dim3(kWave), c->lds_bytes_quad, c->stream, c->d_model, c->d_cfg, mode, t, u_ext,
                       c->d_x0_root, c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_ctrl, c->d_q, c->d_qd, c->d_base, c->d_fr, c->d_cf,
                       c->K == 1 ? c->fb_dof : nullptr, c->K == 1 ? c->fb_root : nullptr);
}
template <class T>
void launch_materialise_scene_t(mppi_ctx *c, float *dof, float *root, float *rb, float *cf) {
    hipLaunchKernelGGL(k_materialise_scene<T>, dim3(c->n_waves), dim3(kWave), 0, c->stream, c->d_model, c->K, c->d_x0_root, c->d_q, c->d_qd, c->d_base,
                       c->d_fr, c->d_cf, dof, root, rb, cf, c->mirror_armed ? c->d_io + kIoDof : nullptr, reinterpret_cast<unsigned *>(c->d_io), c->io_mirror_seq);
}
template <class T>
hipError_t raise_lds_limit(size_t lane_bytes, size_t quad_bytes) {  // lane_bytes == 0: the one-lane kernels are not used
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)quad_bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)quad_bytes);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 8, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)quad_bytes);
    if (e != hipSuccess) return e;
    if constexpr (kHasOctSolve<T>) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 8, 1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)quad_bytes);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 8, 1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)quad_bytes);
        if (e != hipSuccess) return e;
    }
    if constexpr (T::NB <= 4) {  // (+ the helper's accumulator set: bounded by the quad kernel's 16-sample figure)
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 8, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)quad_bytes);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 8, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)quad_bytes);
        if (e != hipSuccess) return e;
    }
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sim_step_scene_quad<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)quad_bytes);
    if (e != hipSuccess || lane_bytes == 0) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rollout_scene<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lane_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sim_step_scene<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lane_bytes);
}

// static __shared__ of the scene kernels (staged start state, fold tickets, step constants, the wave-shared model copy): it
// shares the 160 KiB of a workgroup with the dynamic rows, so the admission check of mppi_create has to count it
template <class T>
size_t static_lds_bytes_scene() {
    size_t mx = 0;
    auto ask = [&](const void *f) {
        hipFuncAttributes a;
        if (hipFuncGetAttributes(&a, f) == hipSuccess && a.sharedSizeBytes > mx) mx = a.sharedSizeBytes;
        else (void)hipGetLastError();
    };
    ask(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 4>));
    ask(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 8>));
    ask(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 8, 1, true>));
    if constexpr (kHasOctSolve<T>) {
        ask(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 8, 1, false, true>));
        ask(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 8, 1, true, true>));
    }
    if constexpr (T::NB <= 4) {
        ask(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 8, 2>));
        ask(reinterpret_cast<const void *>(&k_rollout_scene_quad<T, 8, 2, true>));
    }
    ask(reinterpret_cast<const void *>(&k_sim_step_scene_quad<T>));
    return mx;
}

template <class T>
void launch_rollout_quad_t(mppi_ctx *c) {
    hipLaunchKernelGGL(k_rollout_quad<T>, dim3(c->n_quads), dim3(kWave), 0, c->stream, c->d_model, c->d_cfg, c->d_cost, c->d_x0_dof, c->d_x0_root,
                       c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_S, c->cfg.want_rollouts ? c->d_viz : nullptr, c->d_partials,
                       c->fold ? c->d_fold_ctr : nullptr, c->fold_out, c->wave_clk_on ? c->d_wave_clk : nullptr);
}
// the same rollout with the articulated-body solve in the octet layout (8 lanes per sample, K/8 wavefronts; mppi_oct.hpp)
template <class T>
void launch_rollout_oct_t(mppi_ctx *c) {
    hipLaunchKernelGGL((k_rollout_quad<T, false, 8>), dim3(c->n_quads), dim3(2 * kWave), 0, c->stream, c->d_model, c->d_cfg, c->d_cost, c->d_x0_dof, c->d_x0_root,
                       c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_S, c->cfg.want_rollouts ? c->d_viz : nullptr, c->d_partials,
                       c->fold ? c->d_fold_ctr : nullptr, c->fold_out, c->wave_clk_on ? c->d_wave_clk : nullptr);
}
template <class T>
void launch_rollout_oct_traj_t(mppi_ctx *c) {
    hipLaunchKernelGGL((k_rollout_quad<T, true, 8>), dim3(c->n_quads), dim3(2 * kWave), 0, c->stream, c->d_model, c->d_cfg, c->d_cost_none, c->d_x0_dof, c->d_x0_root,
                       c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_S, (float *)nullptr, c->d_partials,
                       (unsigned *)nullptr, c->fold_out, (unsigned long long *)nullptr, c->d_traj);
}
template <class T>
void launch_rollout_traj_t(mppi_ctx *c) {
    hipLaunchKernelGGL((k_rollout_quad<T, true>), dim3(c->n_quads), dim3(kWave), 0, c->stream, c->d_model, c->d_cfg, c->d_cost_none, c->d_x0_dof, c->d_x0_root,
                       c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_S, (float *)nullptr, c->d_partials,
                       (unsigned *)nullptr, c->fold_out, (unsigned long long *)nullptr, c->d_traj);
}
template <class T>
void launch_materialise_traj_t(mppi_ctx *c, float *dof, float *root, float *rb, float *cf) {
    const size_t HK = (size_t)c->H * c->K;
    hipLaunchKernelGGL(k_materialise<T>, dim3((unsigned)((HK + kWave - 1) / kWave)), dim3(kWave), 0, c->stream, c->d_model, (int)HK, c->d_x0_root, c->d_traj,
                       c->d_traj + (size_t)T::NB * HK, dof, root, rb, cf);
}
template <class T>
void launch_materialise_traj_link_t(mppi_ctx *c, int link, float *out) {
    const size_t HK = (size_t)c->H * c->K;
    hipLaunchKernelGGL(k_materialise_link<T>, dim3((unsigned)((HK + kWave - 1) / kWave)), dim3(kWave), 0, c->stream, c->d_model, (int)HK, c->d_x0_root, c->d_traj,
                       c->d_traj + (size_t)T::NB * HK, link, out);
}
template <class T>
void launch_combine_world_t(mppi_ctx *p, const float *recs, int n, mppi_ctx *w) {
    MailboxArgs mb;
    if (recs == nullptr) {  // (mppi_exchange_update_step_world: the exchange in the same launch)
        mb.wave_recs = p->recs_cur; mb.n_wave = p->n_partials; mb.own = p->d_own_rec; mb.RF = p->RF; mb.rank = p->mb_rank; mb.n = p->mb_n;
        mb.peers = (void *const *)p->d_peers; mb.seq_ctr = p->d_mb_seq; mb.inbox = p->d_inbox; mb.gathered = p->d_gathered; mb.status = p->d_mb_status;
        mb.max_ticks = 200000000ull;
    }
    hipLaunchKernelGGL(k_combine_world<T>, dim3(1), dim3(kCombineWorldThreads), 0, p->stream, p->d_cfg, recs, n, p->d_U, p->d_action, p->d_beta_eta,
                       w->d_model, w->d_x0_root, w->d_q, w->d_qd, p->d_x0_dof, p->use_filter ? p->d_filter : nullptr, mb);
}
template <class T>
void launch_eval_cost_t(mppi_ctx *c, int n, const float *dof, const float *root, const float *rb, const float *cf, float *out) {
    hipLaunchKernelGGL(k_eval_cost<T>, dim3((n + kWave - 1) / kWave), dim3(kWave), 0, c->stream, c->d_model, c->d_cost, n, dof, root, rb, cf, out);
}
template <class T>
void launch_rollout_t(mppi_ctx *c) {
    hipLaunchKernelGGL(k_rollout<T>, dim3(c->n_waves), dim3(kWave), 0, c->stream, c->d_model, c->d_cfg, c->d_cost, c->d_x0_dof, c->d_x0_root,
                       c->d_U, c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_S, c->cfg.want_rollouts ? c->d_viz : nullptr, c->d_partials);
}
template <class T>
void launch_sim_step_t(mppi_ctx *c, int mode, int t, const float *u_ext) {
    hipLaunchKernelGGL(k_sim_step<T>, dim3(c->n_waves), dim3(kWave), 0, c->stream, c->d_model, c->d_cfg, mode, t, u_ext, c->d_x0_root, c->d_U,
                       c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_ctrl, c->d_q, c->d_qd);
}
template <class T>
void launch_sim_step_quad_t(mppi_ctx *c, int mode, int t, const float *u_ext) {
    hipLaunchKernelGGL(k_sim_step_quad<T>, dim3((c->K + 15) / 16), dim3(kWave), 0, c->stream, c->d_model, c->d_cfg, mode, t, u_ext, c->d_x0_root, c->d_U,
                       c->eps_in, c->has_prior ? c->d_prior : nullptr, c->d_du, c->d_ctrl, c->d_q, c->d_qd);
}
template <class T>
void launch_materialise_t(mppi_ctx *c, float *dof, float *root, float *rb, float *cf) {
    hipLaunchKernelGGL(k_materialise<T>, dim3(c->n_waves), dim3(kWave), 0, c->stream, c->d_model, c->K, c->d_x0_root, c->d_q, c->d_qd, dof, root,
                       rb, cf, c->mirror_armed ? c->d_io + kIoDof : nullptr, reinterpret_cast<unsigned *>(c->d_io), c->io_mirror_seq);
}

// The launch-table row of a kinematic tree is filled by TWO translation units (generated: topo_<i>.hip, topo_<i>_scene.hip):
// the contact-free kernels are compiled with the max-ILP machine scheduler (a lone wavefront per SIMD: +3.7 % on the panda
// rollout), the contact-scene kernels with the default one (max-ILP measured -1.3 % / -2.6 % there); it also halves the
// longest compile.
template <class T>
void fill_topo_entry_free(TopoEntry &e) {
    e.nb = T::NB;
    for (int i = 0; i < T::NB; i++) e.parents[i] = T::par[i];
    e.rollout = &launch_rollout_t<T>;
    e.rollout_quad = &launch_rollout_quad_t<T>;
    e.rollout_oct = &launch_rollout_oct_t<T>;
    e.rollout_oct_traj = &launch_rollout_oct_traj_t<T>;
    e.sim_step = &launch_sim_step_t<T>;
    e.sim_step_quad = &launch_sim_step_quad_t<T>;
    e.rollout_traj = &launch_rollout_traj_t<T>;
    e.materialise_traj = &launch_materialise_traj_t<T>;
    e.materialise_traj_link = &launch_materialise_traj_link_t<T>;
    e.materialise = &launch_materialise_t<T>;
    e.combine_world = &launch_combine_world_t<T>;
    e.eval_cost = &launch_eval_cost_t<T>;
}
template <class T>
void fill_topo_entry_scene(TopoEntry &e) {
    e.scene_lds_floats = (size_t)SceneLayout<T>::kCf;
    e.free_slots = kFreeSlots;
    e.rollout_scene = &launch_rollout_scene_t<T>;
    e.rollout_scene_quad = &launch_rollout_scene_quad_t<T, 4>;
    e.rollout_scene_oct = &launch_rollout_scene_quad_t<T, 8>;
    e.rollout_scene_pair = T::NB <= 4 ? &launch_rollout_scene_pair_t<T> : nullptr;
    e.sim_step_scene = &launch_sim_step_scene_t<T>;
    e.sim_step_scene_quad = &launch_sim_step_scene_quad_t<T>;
    e.rollout_scene_traj = &launch_rollout_scene_traj_t<T>;
    e.materialise_scene_traj = &launch_materialise_scene_traj_t<T>;
    e.materialise_scene = &launch_materialise_scene_t<T>;
    e.raise_lds = &raise_lds_limit<T>;
    e.static_lds = &static_lds_bytes_scene<T>;
}

}  // namespace
