// oct_aba_proto.hip - VERDICT r3 item 3: the articulated-body solve of a 7-body chain in today's quad layout (mppi_quad.hpp
// quad_aba: one sample per 4 lanes) and in the octet layout (mppi_oct.hpp oct_aba: angular half of every spatial quantity in one
// quad, linear half in the other, row_ror:8 exchanges) - SAME inputs, ONE lone wavefront per launch, results compared lane by
// lane, time per solve from s_memtime, instruction counts from the ISA (tools/kernel_stats.py on this binary).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -o oct_aba_proto tools/exp/oct_aba_proto.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/mppi_hip.h"
#include "../../mppi-isaac_amd/csrc/mppi_oct.hpp"
#include "../../mppi-isaac_amd/csrc/mppi_quad.hpp"
using namespace mppi;
typedef Topo<-1, 0, 1, 2, 3, 4, 5> Chain7;
constexpr int NB = 7, NS = 8;

template <int LAY>
__global__ __launch_bounds__(64) void k_proto(const DevModel *m, const float *qin, const float *qdin, const float *tauin, float *qdd_out, long long *ticks, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int kModelBytes = (int)((offsetof(DevModel, fr) + 15) / 16 * 16);
    __shared__ __attribute__((aligned(64))) uint4 s_model[kModelBytes / 16];
    __shared__ __attribute__((aligned(64))) DevBody s_lin[NB];
    for (int i = threadIdx.x; i < kModelBytes / 16; i += 64) s_model[i] = reinterpret_cast<const uint4 *>(m)[i];
    __syncthreads();
    LModel &lm = *(LModel *)s_model;
    if (threadIdx.x < NB) {
        DevBody b = ((const DevModel *)m)->b[threadIdx.x];
        b.k1 = oct_lin_view(b.k1);
        s_lin[threadIdx.x] = b;
    }
    __syncthreads();
    const int sample = LAY == 8 ? oct_slot() : (int)(threadIdx.x >> 2) % NS;
    QF q[NB], qd[NB], tau[NB], kdh[NB], qdd[NB];
    JointLimits lim[NB];
    for (int i = 0; i < NB; i++) {
        q[i] = qin[sample * NB + i];
        qd[i] = qdin[sample * NB + i];
        tau[i] = tauin[sample * NB + i];
        kdh[i] = 15.f;
    }
    QPose<Chain7, 0> P;
    QM3 Rb;
    Rb.c[0] = qsel(1.f, 0.f, 0.f); Rb.c[1] = qsel(0.f, 1.f, 0.f); Rb.c[2] = qsel(0.f, 0.f, 1.f);
    P.set_base(Rb, qsel(0.1f, -0.2f, 0.3f));
    quad_fk<Chain7>(lm, q, P);
    const OctLane ol = oct_lane();
    const OctBodies bodies = oct_bodies(&lm.b[0], (const MPPI_LDS_AS DevBody *)s_lin);
    LModel *lp = &lm;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
        LModel &mm = *launder(lp);
        if constexpr (LAY == 8) oct_aba<Chain7>(mm, bodies, ol, P, qd, tau, kdh, qdd, lim);
        else quad_aba<Chain7>(mm, P, qd, tau, kdh, qdd, lim);
        for (int i = 0; i < NB; i++) tau[i] += 1e-7f * qdd[i];   // (the next solve depends on this one)
    }
    long long t1 = clock64();
    for (int i = 0; i < NB; i++) qdd_out[threadIdx.x * NB + i] = qdd[i];
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
#endif
}

static void rot(float ax, float ay, float az, float ang, float *R) {
    float n = std::sqrt(ax * ax + ay * ay + az * az); ax /= n; ay /= n; az /= n;
    float c = std::cos(ang), s = std::sin(ang), C = 1 - c;
    float M[9] = {c + ax * ax * C, ax * ay * C - az * s, ax * az * C + ay * s, ay * ax * C + az * s, c + ay * ay * C, ay * az * C - ax * s,
                  az * ax * C - ay * s, az * ay * C + ax * s, c + az * az * C};
    std::memcpy(R, M, sizeof M);
}

int main() {
    // a 7-body revolute chain with the panda's proportions (z-framed: every joint turns about its local z)
    DevModel hm;
    std::memset(&hm, 0, sizeof hm);
    hm.nb = NB; hm.nl = 1; hm.n_actors = 1; hm.gravity_on = 1; hm.g[2] = -9.8f; hm.nu = NB; hm.substeps = 2; hm.h = 0.025f; hm.kd = 600.f; hm.all_revolute = 1;
    const float off[NB][3] = {{0, 0, 0.333f}, {0, 0, 0}, {0, -0.316f, 0}, {0.0825f, 0, 0}, {-0.0825f, 0.384f, 0}, {0, 0, 0}, {0.088f, 0, 0}};
    const float mass[NB] = {2.98f, 3.0f, 2.33f, 2.37f, 3.42f, 1.44f, 1.2f};
    for (int i = 0; i < NB; i++) {
        DevBody &b = hm.b[i];
        float R[9];
        rot(1, (i % 2) ? -1 : 1, 0.3f * i, (i % 2 ? -1.5707963f : 1.5707963f), R);
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) b.k0.T[4 * r + c] = R[3 * r + c]; b.k0.T[4 * r + 3] = off[i][r]; }
        b.k0.jtype = 0; b.k0.parent = i - 1; b.k0.lower = -INFINITY; b.k0.upper = INFINITY;
        const float hb[3] = {mass[i] * 0.01f * (i + 1), -mass[i] * 0.02f, mass[i] * 0.05f};
        const float I6[6] = {0.02f + 0.003f * i, 0.001f, -0.002f, 0.025f, 0.0015f, 0.012f + 0.001f * i};
        b.k1.set(hb, I6);
        b.k1.m = mass[i]; b.k1.invm = 1.f / mass[i]; b.k1.effort = INFINITY; b.k1.vmax = INFINITY;
        b.cmd.v[i] = 1.f;
    }
    std::vector<float> q(NS * NB), qd(NS * NB), tau(NS * NB);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.f * 2.f - 1.f; };
    for (auto &x : q) x = 2.f * rnd();
    for (auto &x : qd) x = 1.5f * rnd();
    for (auto &x : tau) x = 20.f * rnd();
    DevModel *dm; float *dq, *dqd, *dtau, *dout; long long *dt;
    (void)hipMalloc(&dm, sizeof hm); (void)hipMalloc(&dq, 4 * NS * NB); (void)hipMalloc(&dqd, 4 * NS * NB); (void)hipMalloc(&dtau, 4 * NS * NB);
    (void)hipMalloc(&dout, 4 * 64 * NB); (void)hipMalloc(&dt, 8);
    (void)hipMemcpy(dm, &hm, sizeof hm, hipMemcpyHostToDevice);
    (void)hipMemcpy(dq, q.data(), 4 * NS * NB, hipMemcpyHostToDevice);
    (void)hipMemcpy(dqd, qd.data(), 4 * NS * NB, hipMemcpyHostToDevice);
    (void)hipMemcpy(dtau, tau.data(), 4 * NS * NB, hipMemcpyHostToDevice);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const double clk_ratio = p.clockRate / 1000.0 / 100.0;   // s_memtime: 100 MHz
    std::vector<float> o4(64 * NB), o8(64 * NB);
    double per[2] = {0, 0};
    for (int lay = 0; lay < 2; lay++) {
        long long best = 1LL << 62;
        const int iters = 400;
        for (int rep = 0; rep < 6; rep++) {
            if (lay == 0) hipLaunchKernelGGL(k_proto<4>, dim3(1), dim3(64), 0, 0, dm, dq, dqd, dtau, dout, dt, rep == 0 ? 1 : iters);
            else hipLaunchKernelGGL(k_proto<8>, dim3(1), dim3(64), 0, 0, dm, dq, dqd, dtau, dout, dt, rep == 0 ? 1 : iters);
            (void)hipDeviceSynchronize();
            long long t; (void)hipMemcpy(&t, dt, 8, hipMemcpyDeviceToHost);
            if (rep == 0) (void)hipMemcpy(lay == 0 ? o4.data() : o8.data(), dout, 4 * 64 * NB, hipMemcpyDeviceToHost);   // ONE solve: the result to compare
            else if (t < best) best = t;
        }
        per[lay] = (double)best / iters;
        printf("%s layout: %8.2f ticks of s_memtime per solve of the 7-body chain = %7.0f shader cycles (%d solves, lone wavefront)\n",
               lay == 0 ? "quad (4 lanes/sample) " : "octet (8 lanes/sample)", per[lay], per[lay] * clk_ratio, iters);
    }
    // same inputs -> same accelerations: sample s sits in lanes 4s.. of the quad layout, in slot s of the octet layout
    double worst = 0, scale = 0;
    bool lanes_equal = true;
    for (int l = 0; l < 64; l++) {
        const int row = l >> 4, qir = (l >> 2) & 3, slot = row * 2 + (qir & 1);
        for (int i = 0; i < NB; i++) {
            const double a = o4[(4 * slot) * NB + i], b = o8[l * NB + i];
            worst = std::fmax(worst, std::fabs(a - b));
            scale = std::fmax(scale, std::fabs(a));
            const int l0 = (row * 16 + (qir & 1) * 4);   // first lane of the sample's angular quad
            if (o8[l * NB + i] != o8[l0 * NB + i]) lanes_equal = false;
        }
    }
    printf("qdd octet vs quad: max abs difference %.3e (largest |qdd| %.1f), the eight lanes of every sample bit-identical: %s\n", worst, scale, lanes_equal ? "yes" : "NO");
    printf("octet / quad time per solve: %.3f  (%+.1f %%)\n", per[1] / per[0], 100.0 * (per[1] / per[0] - 1.0));
    return (worst <= 2e-4 * scale && lanes_equal) ? 0 : 1;
}
