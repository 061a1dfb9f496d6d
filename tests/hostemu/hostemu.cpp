// hostemu.cpp - TEST-ONLY host build of the per-sample device functions (csrc/mppi_device.hpp).
// g++ compiles the very same templates the GPU kernels instantiate, so the fp32 world-frame
// arithmetic can be compared with the oracle on a machine without a GPU (pytest -m "not gpu").
// Nothing in the product path loads this library; the wave-level reductions and launches in
// mppi_hip.hip are only exercised by the -m gpu tests.
#include <string>
#include <vector>

#include "../../mppi-isaac_amd/csrc/mppi_pack.hpp"
#include "../../mppi-isaac_amd/csrc/mppi_scene.hpp"
#include "../../mppi-isaac_amd/csrc/mppi_quad.hpp"
#include "../../mppi-isaac_amd/csrc/mppi_scene_quad.hpp"

using namespace mppi;

// The file is compiled three times (HOSTEMU_PART = 1, 2, 3: the Makefile builds the parts in parallel - one translation unit
// with every kinematic tree in every entry point took over two minutes) and linked into one library.
#ifndef HOSTEMU_PART
#error "compile with -DHOSTEMU_PART=1|2|3 (tests/hostemu/Makefile)"
#endif
#if HOSTEMU_PART == 3
int g_scene_split = 1;  // 4 / 8: contact feature points dealt over an emulated quad / octet (kSplitEmulate)
#else
extern int g_scene_split;
#endif

extern "C" {

#if HOSTEMU_PART == 3
void emu_set_scene_split(int n) { g_scene_split = n; }
#endif

#if HOSTEMU_PART == 1
int emu_rollout(const mppi_model_t *model, const mppi_config_t *cfg, const mppi_cost_t *cost, const float *dof0, const float *root0,
                const float *U, const float *eps, const float *prior, float *S, float *du, float *viz) {
    DevModel m; DevCfg c; DevCost k; std::string err;
    if (!pack_model(*model, m, err) || !pack_config(*cfg, c, err) || !pack_cost(*cost, m, k, err)) return -1;
    int parents[MPPI_MAX_BODIES];
    for (int i = 0; i < m.nb; i++) parents[i] = m.b[i].k0.parent;
    bool ok = dispatch_topology(m.nb, parents, [&](auto topo) {
        using T = decltype(topo);
        std::vector<float> v(viz ? (size_t)c.H * 3 * c.K : 0);
        if (is_scene(m)) {
            std::vector<float> lmem(scene_row_floats<T>(m));
            LMem L{lmem.data(), 1};
            L.lp = lmem.data() + scene_light_base<T>(m);   // (records of the light bodies' pairs: the tail of the rows)
            L.lstride = 1;
            if (m.n_light_pairs != 0) light_region_reset(L);
            // as the rollout kernels do: start state relative to the robot's start position (mppi_scene.hpp root_relative)
            std::vector<float> rel(13 * m.n_actors);
            root_origin(m, root0, L.ox, L.oy);
            for (int j = 0; j < 13 * m.n_actors; j++) rel[j] = root_relative_entry(root0, j, L.ox, L.oy);
            for (int s = 0; s < c.K; s++)
                S[s] = g_scene_split > 1 ? rollout_scene<T, kSplitEmulate>(m, m, c, k, dof0, rel.data(), U, eps, prior, du, viz ? v.data() : nullptr, s, L, Split{0, g_scene_split})
                                         : rollout_scene<T>(m, m, c, k, dof0, rel.data(), U, eps, prior, du, viz ? v.data() : nullptr, s, L);
        } else
        for (int s = 0; s < c.K; s++) S[s] = rollout_sample<T>(m, c, k, dof0, root0, U, eps, prior, du, viz ? v.data() : nullptr, s);
        if (viz)  // device layout [H][3][K] -> reference layout [H][K][3]
            for (int t = 0; t < c.H; t++)
                for (int j = 0; j < 3; j++)
                    for (int s = 0; s < c.K; s++) viz[((size_t)t * c.K + s) * 3 + j] = v[((size_t)t * 3 + j) * c.K + s];
    });
    return ok ? 0 : -3;
}

#endif  // part 1
#if HOSTEMU_PART == 2
// quad-parallel rollout (csrc/mppi_quad.hpp) with the 4-lane quad emulated as a 4-float struct
int emu_rollout_quad(const mppi_model_t *model, const mppi_config_t *cfg, const mppi_cost_t *cost, const float *dof0, const float *root0,
                     const float *U, const float *eps, const float *prior, float *S, float *du, float *viz) {
    DevModel m; DevCfg c; DevCost k; std::string err;
    if (!pack_model(*model, m, err) || !pack_config(*cfg, c, err) || !pack_cost(*cost, m, k, err)) return -1;
    if (is_scene(m)) return -4;
    int parents[MPPI_MAX_BODIES];
    for (int i = 0; i < m.nb; i++) parents[i] = m.b[i].k0.parent;
    bool ok = dispatch_topology(m.nb, parents, [&](auto topo) {
        using T = decltype(topo);
        std::vector<float> v(viz ? (size_t)c.H * 3 * c.K : 0);
        StepConsts sc;
        std::memset(&sc, 0, sizeof sc);
        for (int j = 0; j < step_const_count(c); j++) reinterpret_cast<float *>(&sc)[j] = step_const_entry(c, k, root0, U, j);
        for (int s = 0; s < c.K; s++) {
            QF r = m.all_revolute ? quad_rollout<T, 0>(m, c, k, sc, dof0, root0, eps, prior, du, viz ? v.data() : nullptr, s, true, 0, true)
                                  : quad_rollout<T, -1>(m, c, k, sc, dof0, root0, eps, prior, du, viz ? v.data() : nullptr, s, true, 0, true);
            S[s] = r.v[0];
            if (r.v[1] != r.v[0] || r.v[2] != r.v[0] || r.v[3] != r.v[0]) S[s] = NAN;  // replicated scalars must agree across the quad
        }
        if (viz)
            for (int t = 0; t < c.H; t++)
                for (int j = 0; j < 3; j++)
                    for (int s = 0; s < c.K; s++) viz[((size_t)t * c.K + s) * 3 + j] = v[((size_t)t * 3 + j) * c.K + s];
    });
    return ok ? 0 : -3;
}

int emu_step(const mppi_model_t *model, const float *root, float *q, float *qd, const float *u) {
    DevModel m; std::string err;
    if (!pack_model(*model, m, err)) return -1;
    int parents[MPPI_MAX_BODIES];
    for (int i = 0; i < m.nb; i++) parents[i] = m.b[i].k0.parent;
    bool ok = dispatch_topology(m.nb, parents, [&](auto topo) {
        using T = decltype(topo);
        float target[MPPI_MAX_BODIES], uu[kMaxNu] = {0};
        for (int c = 0; c < m.nu; c++) uu[c] = u[c];
        cmd_map<T>(m, uu, target);
        step<T>(m, root, q, qd, target);
    });
    return ok ? 0 : -3;
}

int emu_rigid_body_state(const mppi_model_t *model, const float *root, const float *q, const float *qd, float *rb, float *cf) {
    DevModel m; std::string err;
    if (!pack_model(*model, m, err)) return -1;
    int parents[MPPI_MAX_BODIES];
    for (int i = 0; i < m.nb; i++) parents[i] = m.b[i].k0.parent;
    bool ok = dispatch_topology(m.nb, parents, [&](auto topo) {
        using T = decltype(topo);
        rigid_body_state<T>(m, root, q, qd, rb, cf);
    });
    return ok ? 0 : -3;
}

#endif  // part 2
#if HOSTEMU_PART == 3
// one dt step of a contact scene: dof [2n] and root [A][13] are updated in place; rb/cf = reference-layout rows
int emu_scene_step_g(const mppi_model_t *model, float *dof, float *root, const float *u, float *rb, float *cf, int sample_id) {
    DevModel m; std::string err;
    if (!pack_model(*model, m, err)) return -1;
    int parents[MPPI_MAX_BODIES];
    for (int i = 0; i < m.nb; i++) parents[i] = m.b[i].k0.parent;
    bool ok = dispatch_topology(m.nb, parents, [&](auto topo) {
        using T = decltype(topo);
        std::vector<float> lmem(scene_row_floats<T>(m));
        LMem L{lmem.data(), 1};
        L.lp = lmem.data() + scene_light_base<T>(m);
        L.lstride = 1;
        if (m.n_light_pairs != 0) light_region_reset(L);
        SceneState<T> s;
        scene_init<T>(m, dof, root, s, sample_id, L);
        float target[MPPI_MAX_BODIES + 1], uu[kMaxNu] = {0};
        for (int c = 0; c < m.nu; c++) uu[c] = u[c];
        cmd_map<T>(m, uu, target);
        if (g_scene_split > 1) {
            shape_cache_update<T>(m, root, L, Split{0, 1}, true);
            step_scene_quad<T, kSplitEmulate>(m, m, root, s, target, L, Split{0, g_scene_split});
        }
        else step_scene<T>(m, root, s, target, L);
        for (int i = 0; i < T::NB; i++) { dof[2 * i] = s.q[i]; dof[2 * i + 1] = s.qd[i]; }
        std::vector<float> rootn(13 * m.n_actors);
        scene_materialise<T>(m, root, s, lmem.data() + SceneLayout<T>::kCf, rootn.data(), rb, cf);
        for (int j = 0; j < 13 * m.n_actors; j++) root[j] = rootn[j];
    });
    return ok ? 0 : -3;
}

int emu_scene_step(const mppi_model_t *model, float *dof, float *root, const float *u, float *rb, float *cf) {
    return emu_scene_step_g(model, dof, root, u, rb, cf, 0);
}

#endif  // part 3
#if HOSTEMU_PART == 2
float emu_cost(const mppi_model_t *model, const mppi_cost_t *cost, const float *root, const float *q, const float *qd) {
    DevModel m; DevCost k; std::string err;
    if (!pack_model(*model, m, err) || !pack_cost(*cost, m, k, err)) return -1e30f;
    int parents[MPPI_MAX_BODIES];
    for (int i = 0; i < m.nb; i++) parents[i] = m.b[i].k0.parent;
    float out = -1e30f;
    dispatch_topology(m.nb, parents, [&](auto topo) {
        using T = decltype(topo);
        out = stage_cost<T>(m, k, root, q, qd);
    });
    return out;
}
#endif  // part 2 (cost)
}
