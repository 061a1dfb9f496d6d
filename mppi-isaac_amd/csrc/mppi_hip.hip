// mppi_hip.hip - C-ABI (include/mppi_hip.h) of the MPPI rollout backend.  Kernels, context and the
// per-topology launch table live in mppi_kernels.hpp; the compile-time kinematic trees are instantiated in
// the generated topo_<i>.hip units and found through topo_table.inc.
#include <atomic>
#include <chrono>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>
#include <unordered_set>

#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "mppi_kernels.hpp"
#include "topo_table.inc"  // generated: extern "C" const TopoEntry *mppi_topo_entry_<i>(); kTopoEntries[]

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            (void)hipGetLastError(); /* reported here: the next launch_check() must not find it again */      \
            return fail(MPPI_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));                        \
        }                                                                                                     \
    } while (0)

struct EvScope {  // optional hipEvent bracket around one launch (profiling mode)
    mppi_ctx *c;
    int which;
    hipEvent_t stop = nullptr;
    EvScope(mppi_ctx *c_, int w) : c(c_), which(w) {
        if (!c->profiling) return;
        if (c->ev_seen[which]++ % (size_t)c->profile_period != 0) return;  // every n-th launch of this kernel
        auto &v = c->ev[which];
        if (c->ev_used[which] >= 8192) return;  // bounded: a long profiled run keeps its first 8192 brackets per kernel
        if (c->ev_used[which] == v.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            v.emplace_back(a, b);
        }
        auto &p = v[c->ev_used[which]++];
        (void)hipEventRecord(p.first, c->stream);
        stop = p.second;
    }
    ~EvScope() {
        if (stop) (void)hipEventRecord(stop, c->stream);
    }
};

template <class P>
int dev_alloc(P **p, size_t bytes) {
    HIP_TRY(hipMalloc((void **)p, bytes ? bytes : 4));
    HIP_TRY(hipMemset(*p, 0, bytes ? bytes : 4));
    return MPPI_OK;
}
#define ALLOC_TRY(p, bytes)                 \
    do {                                    \
        int rc_ = dev_alloc(&(p), (bytes)); \
        if (rc_) return rc_;                \
    } while (0)

// live handles: a destroyed (stale) or foreign pointer is reported instead of dereferenced
std::mutex g_live_mu;
std::unordered_set<const mppi_ctx *> g_live;
int check_ctx(const mppi_ctx *c) {
    if (!c) return fail(MPPI_EINVAL, "null context");
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (g_live.find(c) == g_live.end()) return fail(MPPI_EINVAL, "stale or foreign context handle (destroyed by mppi_destroy?)");
    return MPPI_OK;
}
#define CTX_TRY(c)            \
    do {                      \
        int rc_ = check_ctx(c); \
        if (rc_) return rc_;  \
        hipError_t e_ = hipSetDevice((c)->device); \
        if (e_ != hipSuccess) return fail(MPPI_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e_)); \
    } while (0)

// device / pinned allocations and initial uploads of a new context; any failure leaves a partially filled context that
// release_ctx() disposes of
int create_buffers(mppi_ctx *c, const mppi_config_t *cfg) {
    c->n = c->hm.nb; c->A = c->hm.n_actors; c->B = c->hm.n_rb; c->K = cfg->num_samples; c->H = cfg->horizon; c->nu = cfg->nu;
    c->HN = c->H * c->nu; c->RF = 2 + c->HN; c->n_waves = (c->K + kWave - 1) / kWave;
    {
        // samples per workgroup (= per record) of the rollout kernel that shares lanes: 16, contact scenes in the octet layout 8
        // (the contact-free octet kernel runs two wavefronts of eight samples per workgroup)
        const int spw = (c->lanes_per_sample == 8 && c->scene) ? 8 : 16;
        c->n_quads = (c->K + spw - 1) / spw;
    }
    const size_t K = c->K;
    ALLOC_TRY(c->d_model, sizeof(DevModel));
    ALLOC_TRY(c->d_cfg, sizeof(DevCfg));
    ALLOC_TRY(c->d_cost, sizeof(DevCost));
    ALLOC_TRY(c->d_x0_dof, sizeof(float) * 2 * c->n);
    ALLOC_TRY(c->d_x0_root, sizeof(float) * 13 * c->A);
    ALLOC_TRY(c->d_U, sizeof(float) * c->HN);
    ALLOC_TRY(c->d_eps, sizeof(float) * c->HN * K);
    ALLOC_TRY(c->d_du, sizeof(float) * c->HN * K);
    ALLOC_TRY(c->d_S, sizeof(float) * K);
    ALLOC_TRY(c->d_prior, sizeof(float) * c->HN);
    ALLOC_TRY(c->d_viz, sizeof(float) * (cfg->want_rollouts ? (size_t)c->H * K * 3 : 1));
    ALLOC_TRY(c->d_partials, sizeof(float) * (size_t)c->n_quads * c->RF);
    ALLOC_TRY(c->d_record, sizeof(float) * c->RF);
    ALLOC_TRY(c->d_fold, sizeof(float) * kFoldGroups * c->RF);
    ALLOC_TRY(c->d_fold_ctr, sizeof(unsigned) * kFoldGroups);
    c->fold_out = c->d_fold;
    ALLOC_TRY(c->d_action, sizeof(float) * c->nu);
    ALLOC_TRY(c->d_beta_eta, sizeof(float) * 2);
    ALLOC_TRY(c->d_q, sizeof(float) * c->n * K);
    ALLOC_TRY(c->d_qd, sizeof(float) * c->n * K);
    ALLOC_TRY(c->d_ctrl, sizeof(float) * K);
    ALLOC_TRY(c->d_base, sizeof(float) * 13 * (size_t)c->hm.n_bases * K);   // (one block of 13 rows per moving base of the env)
    ALLOC_TRY(c->d_fr, sizeof(float) * c->free_slots * 13 * K);
    ALLOC_TRY(c->d_cf, sizeof(float) * 3 * c->B * K);
    ALLOC_TRY(c->d_filter, sizeof(float) * c->H * c->H);
    ALLOC_TRY(c->d_basis, sizeof(double) * MPPI_MAX_H * MPPI_MAX_KNOTS);
    ALLOC_TRY(c->d_sigma, sizeof(double) * 2 * MPPI_MAX_NU);  // sqrt(sigma_diag) | noise_mu
    c->eps_in = c->d_eps;
    double sig[2 * MPPI_MAX_NU] = {0};
    for (int j = 0; j < c->nu; j++) {
        sig[j] = std::sqrt(cfg->noise_sigma_diag[j]);
        sig[MPPI_MAX_NU + j] = cfg->noise_mu[j];
    }
    HIP_TRY(hipMemcpy(c->d_model, &c->hm, sizeof(DevModel), hipMemcpyHostToDevice));
    // the update kernels also store the action into mapped pinned host memory: mppi_get_action is then a stream
    // synchronise + a host read instead of a D2H copy operation
    HIP_TRY(hipHostMalloc((void **)&c->h_action, sizeof(float) * 32, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->h_action, 0, sizeof(float) * 32);
    HIP_TRY(hipHostGetDevicePointer((void **)&c->hc.action_mirror, c->h_action, 0));
    c->hc.seq_host = reinterpret_cast<unsigned *>(c->hc.action_mirror + 16);
    HIP_TRY(hipHostMalloc((void **)&c->h_io, sizeof(float) * kIoFloats, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->h_io, 0, sizeof(float) * kIoFloats);
    HIP_TRY(hipHostGetDevicePointer((void **)&c->d_io, c->h_io, 0));
    ALLOC_TRY(c->d_seq, sizeof(unsigned));
    c->hc.seq_dev = c->d_seq;
    HIP_TRY(hipMemcpy(c->d_cfg, &c->hc, sizeof(DevCfg), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_cost, &c->hk, sizeof(DevCost), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_basis, cfg->spline_basis, sizeof(double) * MPPI_MAX_H * MPPI_MAX_KNOTS, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_sigma, sig, sizeof sig, hipMemcpyHostToDevice));
    std::vector<float> U0(c->HN, (float)cfg->u_init);
    HIP_TRY(hipMemcpy(c->d_U, U0.data(), sizeof(float) * c->HN, hipMemcpyHostToDevice));
    return MPPI_OK;
}

// frees everything a context owns (null-safe: also used for a context whose creation failed half way)
void release_ctx(mppi_ctx *c) {
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    void *bufs[] = {c->d_model, c->d_cfg, c->d_cost, c->d_x0_dof, c->d_x0_root, c->d_U, c->d_eps, c->d_du, c->d_S, c->d_prior, c->d_viz,
                    c->d_partials, c->d_record, c->d_action, c->d_beta_eta, c->d_q, c->d_qd, c->d_ctrl, c->d_basis, c->d_sigma, c->d_base, c->d_fr, c->d_cf, c->d_filter,
                    c->d_seq, c->d_fold, c->d_fold_ctr, c->d_wave_clk, c->d_traj, c->d_cost_none, c->d_inbox, c->d_peers, c->d_mb_seq,
                    c->d_gathered, c->d_own_rec};  // (d_mb_status lives in the mapped host block h_action)
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    for (void *p : c->ipc_opened) (void)hipIpcCloseMemHandle(p);
    if (c->h_action) (void)hipHostFree(c->h_action);
    if (c->h_io) (void)hipHostFree(c->h_io);
    for (auto &v : c->ev)
        for (auto &p : v) {
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
    delete c;
}

int launch_check() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MPPI_EHIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return MPPI_OK;
}

// ---- host hand-over without copy operations (ABI 8: the bytes API of the reference hands ONE env state in and takes one
// action out per control iteration, < 1 KB either way - a hipMemcpyAsync + synchronise per tensor cost more than the data) ----
// the state travels as KERNEL ARGUMENTS of a one-wavefront launch: no copy operation, no synchronisation, and the caller's
// buffers are free again when the call returns (the runtime copies kernel arguments at launch)
struct X0Args {
    float v[2 * MPPI_MAX_BODIES + 13 * MPPI_MAX_ACTORS];
};
__global__ void k_set_x0(X0Args a, int n_dof, int n_root, float *__restrict__ x0_dof, float *__restrict__ x0_root) {
    for (int i = threadIdx.x; i < n_dof; i += blockDim.x) x0_dof[i] = a.v[i];
    for (int i = threadIdx.x; i < n_root; i += blockDim.x) x0_root[i] = a.v[2 * MPPI_MAX_BODIES + i];
}
}  // namespace

// ------------------------------------------------------------------------------ kinematic trees built on demand
// The rollout kernels are templates over the kinematic tree (every per-body array index static: mppi_device.hpp), so a tree has to
// be INSTANTIATED before it can run.  The library ships the instantiations of the robots under assets/compiled/; any other tree -
// what the reference gets from gym.load_asset for whatever URDF the actor YAML names (isaacgym_utils.py:14-29,
// isaacgym_wrapper.py:429-447) - is built HERE at mppi_create: the two generated units of that tree (exactly what
// __graft_entry__.generate_sources writes per tree) are compiled with hipcc for gfx950 into a small plugin library, cached on
// disk under a key of the tree, the build flags and the contents of the kernel headers, dlopen'ed and appended to the launch
// table.  The first planner of a new robot pays the compile (tens of seconds, like a convex decomposition in the reference's
// asset import); every later one finds the plugin in the cache.  MPPI_JIT=0 switches this off (unknown trees are then refused),
// MPPI_JIT_CACHE names the cache directory, HIPCC the compiler.
extern "C" int mppi_create(const mppi_model_t *, const mppi_config_t *, int, mppi_ctx_t **);
extern "C" char **environ;
namespace {

std::mutex g_plugin_mu;
std::vector<const TopoEntry *> g_plugins;   // entries of the plugins loaded so far (never unloaded)
std::string g_jit_log;                        // what the last on-demand build did / what the running one is doing (mppi_jit_info)
bool g_jit_building = false;                  // a build is running: mppi_jit_info adds the seconds it has taken so far
std::chrono::steady_clock::time_point g_jit_t0;
std::mutex g_jit_mu;                          // one on-demand build at a time per process (two threads asking for the same new tree)
std::atomic<unsigned> g_jit_serial{0};        // build temporaries: pid + serial

const char *const kJitFlags[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"};   // = __graft_entry__.HIPCC_FLAGS
const char *const kJitFreeFlags[] = {"-mllvm", "-amdgpu-sched-strategy=max-ilp"};                                     // = ILP_FLAGS (contact-free units)
constexpr int kDppLeadWaitFrom = 9;                                                                                    // = DPP_LEAD_WAIT_FROM

bool entry_matches(const TopoEntry *e, int nb, const int *parents, bool scene, int n_free) {
    if (e->nb != nb) return false;
    for (int i = 0; i < nb; i++)
        if (e->parents[i] != parents[i]) return false;
    if (!e->rollout) return false;
    if (scene && (!e->rollout_scene || e->free_slots < n_free)) return false;
    return true;
}
const TopoEntry *find_entry(int nb, const int *parents, bool scene, int n_free) {
    for (const TopoEntryFn fn : kTopoEntries)
        if (entry_matches(fn(), nb, parents, scene, n_free)) return fn();
    std::lock_guard<std::mutex> lk(g_plugin_mu);
    for (const TopoEntry *e : g_plugins)
        if (entry_matches(e, nb, parents, scene, n_free)) return e;
    return nullptr;
}

std::string lib_dir() {   // directory of this library = the directory of the kernel headers (csrc/)
    Dl_info info;
    if (dladdr((void *)&mppi_create, &info) == 0 || !info.dli_fname) return "";
    std::string p = info.dli_fname;
    const size_t k = p.rfind('/');
    return k == std::string::npos ? "." : p.substr(0, k);
}
bool file_exists(const std::string &p) {
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
// the cache directory itself is created 0700 (its parents 0755)
bool mkdir_p(const std::string &dir) {
    for (size_t i = 1; i <= dir.size(); i++)
        if (i == dir.size() || dir[i] == '/') {
            const std::string d = dir.substr(0, i);
            if (mkdir(d.c_str(), i == dir.size() ? 0700 : 0755) != 0 && errno != EEXIST) return false;
        }
    return true;
}
// code is loaded from the cache: directory and plugin must belong to this process's user and be writable by nobody else - a plugin's
// file name is computable, and whoever can write there decides what mppi_create executes
bool owned_and_private(const std::string &path, bool is_dir, std::string &why) {
    struct stat st;
    if (stat(path.c_str(), &st) != 0) { why = path + ": " + std::strerror(errno); return false; }
    if (is_dir ? !S_ISDIR(st.st_mode) : !S_ISREG(st.st_mode)) { why = path + " is not a " + (is_dir ? "directory" : "regular file"); return false; }
    if (st.st_uid != geteuid()) { why = path + " belongs to uid " + std::to_string((long)st.st_uid) + ", this process runs as " + std::to_string((long)geteuid()); return false; }
    if (st.st_mode & (S_IWGRP | S_IWOTH)) {
        char m[8];
        std::snprintf(m, sizeof m, "%04o", (unsigned)(st.st_mode & 07777));
        why = path + " is writable by group or others (mode " + m + ")";
        return false;
    }
    return true;
}
// FNV-1a over the kernel headers and the ABI header: a plugin is only ever loaded next to the sources it was built from
bool source_key(const std::string &dir, uint64_t &key, std::string &err) {
    const char *files[] = {"mppi_kernels.hpp", "mppi_device.hpp", "mppi_quad.hpp", "mppi_oct.hpp", "mppi_scene.hpp", "mppi_scene_quad.hpp", "mppi_scene_oct.hpp",
                           "mppi_pack.hpp", "topologies.inc", "../../include/mppi_hip.h"};
    uint64_t h = 1469598103934665603ull;
    for (const char *f : files) {
        std::ifstream in(dir + "/" + f, std::ios::binary);
        if (!in) { err = "kernel sources not found next to the library (" + dir + "/" + f + ")"; return false; }
        char buf[1 << 16];
        while (in.read(buf, sizeof buf) || in.gcount() > 0) {
            for (std::streamsize i = 0; i < in.gcount(); i++) h = (h ^ (unsigned char)buf[i]) * 1099511628211ull;
            if (!in) break;
        }
    }
    for (const char *f : kJitFlags)
        for (const char *q = f; *q; q++) h = (h ^ (unsigned char)*q) * 1099511628211ull;
    key = h;
    return true;
}
std::string cache_dir() {
    if (const char *e = std::getenv("MPPI_JIT_CACHE")) return e;
    if (const char *e = std::getenv("XDG_CACHE_HOME")) return std::string(e) + "/mppi_hip";
    if (const char *e = std::getenv("HOME")) return std::string(e) + "/.cache/mppi_hip";
    return "/tmp/mppi_hip_" + std::to_string((long)getuid());
}
std::string find_hipcc() {
    if (const char *e = std::getenv("HIPCC")) return e;
    for (const char *p : {"/opt/rocm/bin/hipcc", "/usr/bin/hipcc", "/usr/local/bin/hipcc"})
        if (access(p, X_OK) == 0) return p;
    return "";
}
// runs one compiler process, stderr + stdout into `log`; returns its pid (-1: could not start)
pid_t spawn_to_log(const std::vector<std::string> &argv, const std::string &log) {
    std::vector<char *> av;
    for (const std::string &a : argv) av.push_back(const_cast<char *>(a.c_str()));
    av.push_back(nullptr);
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    posix_spawn_file_actions_adddup2(&fa, 1, 2);
    pid_t pid = -1;
    const int rc = posix_spawn(&pid, av[0], &fa, nullptr, av.data(), environ);
    posix_spawn_file_actions_destroy(&fa);
    return rc == 0 ? pid : -1;
}
bool wait_ok(pid_t pid) {
    int st = 0;
    while (waitpid(pid, &st, 0) < 0)
        if (errno != EINTR) return false;
    return WIFEXITED(st) && WEXITSTATUS(st) == 0;
}
std::string tail_of(const std::string &path, size_t n = 1500) {
    std::ifstream in(path);
    std::stringstream ss;
    ss << in.rdbuf();
    std::string t = ss.str();
    return t.size() > n ? t.substr(t.size() - n) : t;
}

// builds (or finds in the cache) and loads the plugin of one kinematic tree; `with_scene`: the contact-scene kernels as well
// (the long compile), `free_slots`: free-actor slots of those kernels
const TopoEntry *jit_topology(int nb, const int *parents, bool with_scene, int free_slots, std::string &err, bool retried = false) {
    const char *sw = std::getenv("MPPI_JIT");
    if (sw && std::string(sw) == "0") { err = "on-demand builds are switched off (MPPI_JIT=0)"; return nullptr; }
    const std::string dir = lib_dir();
    uint64_t key = 0;
    if (dir.empty() || !source_key(dir, key, err)) { if (err.empty()) err = "cannot locate the library on disk"; return nullptr; }
    std::string name = "topo";
    for (int i = 0; i < nb; i++) name += (parents[i] < 0 ? "_m" : "_") + std::to_string(parents[i] < 0 ? -parents[i] : parents[i]);
    char kbuf[32];
    std::snprintf(kbuf, sizeof kbuf, "%016llx", (unsigned long long)key);
    name += std::string(with_scene ? "_scene" + std::to_string(free_slots) : "_free") + "_abi" + std::to_string(MPPI_ABI_VERSION) + "_" + kbuf;
    const std::string cdir = cache_dir();
    if (!mkdir_p(cdir)) { err = "cannot create the plugin cache directory " + cdir + " (MPPI_JIT_CACHE)"; return nullptr; }
    std::string why_not;
    if (!owned_and_private(cdir, true, why_not)) { err = "refusing the plugin cache directory: " + why_not + " (MPPI_JIT_CACHE names another one)"; return nullptr; }
    const std::string so = cdir + "/" + name + ".so";
    const auto t0 = std::chrono::steady_clock::now();
    bool built = false;
    std::unique_lock<std::mutex> build_lock(g_jit_mu, std::defer_lock);
    if (!retried) build_lock.lock();
    if (!file_exists(so)) {
        const std::string hipcc = find_hipcc();
        if (hipcc.empty()) { err = "no hipcc found (HIPCC, /opt/rocm/bin/hipcc): cannot build the kernels of this tree"; return nullptr; }
        std::string args;
        for (int i = 0; i < nb; i++) args += (i ? ", " : "") + std::to_string(parents[i]);
        const std::string tmp = cdir + "/" + name + "." + std::to_string((long)getpid()) + "_" + std::to_string(g_jit_serial.fetch_add(1));
        const std::string inc = "#include \"" + dir + "/mppi_kernels.hpp\"\n";
        {
            std::ofstream f(tmp + "_free.hip");
            f << "// GENERATED by libmppi_hip.so (jit_topology): contact-free kernels of the kinematic tree [" << args << "]\n"
              << (nb > kDppLeadWaitFrom ? "#define MPPI_DPP_LEAD_WAIT 1\n" : "") << inc
              << (with_scene ? "extern \"C\" void mppi_plugin_fill_scene(mppi::TopoEntry *e);\n" : "")
              << "extern \"C\" const mppi::TopoEntry *mppi_plugin_entry() {\n    static const mppi::TopoEntry e = [] {\n        mppi::TopoEntry x{};\n"
              << "        fill_topo_entry_free<mppi::Topo<" << args << ">>(x);\n"
              << (with_scene ? "        mppi_plugin_fill_scene(&x);\n" : "") << "        return x;\n    }();\n    return &e;\n}\n"
              << "extern \"C\" void mppi_plugin_sizes(size_t *out) { out[0] = sizeof(mppi_ctx); out[1] = sizeof(mppi::DevModel); out[2] = sizeof(mppi::TopoEntry); out[3] = sizeof(mppi::DevCfg); }\n";
        }
        std::vector<std::pair<pid_t, std::string>> jobs;
        std::vector<std::string> objs = {tmp + "_free.o"};
        std::vector<std::string> cmd = {hipcc};
        for (const char *f : kJitFlags) cmd.push_back(f);
        std::vector<std::string> cfree = cmd;
        for (const char *f : kJitFreeFlags) cfree.push_back(f);
        cfree.insert(cfree.end(), {"-c", "-o", tmp + "_free.o", tmp + "_free.hip"});
        std::fprintf(stderr, "[mppi_hip] building the kernels of kinematic tree [%s]%s with %s (one-off: cached as %s)\n", args.c_str(),
                     with_scene ? " incl. the contact-scene kernels" : "", hipcc.c_str(), so.c_str());
        {   // (progress: mppi_jit_info from another thread says what this call is waiting for)
            std::lock_guard<std::mutex> lk(g_plugin_mu);
            g_jit_log = "building " + so + " (tree [" + args + "]" + (with_scene ? ", with the contact-scene kernels" : "") + ")";
            g_jit_building = true;
            g_jit_t0 = t0;
        }
        struct Done { ~Done() { std::lock_guard<std::mutex> lk(g_plugin_mu); g_jit_building = false; } } done_guard;
        jobs.emplace_back(spawn_to_log(cfree, tmp + "_free.log"), tmp + "_free.log");
        if (with_scene) {
            std::ofstream f(tmp + "_scene.hip");
            f << "// GENERATED by libmppi_hip.so (jit_topology): contact-scene kernels of the kinematic tree [" << args << "]\n#define MPPI_DPP_LEAD_WAIT 1\n"
              << inc << "extern \"C\" void mppi_plugin_fill_scene(mppi::TopoEntry *e) { fill_topo_entry_scene<mppi::Topo<" << args << ">>(*e); }\n";
            f.close();
            std::vector<std::string> cs = cmd;
            cs.insert(cs.end(), {"-DMPPI_FREE_SLOTS=" + std::to_string(free_slots), "-c", "-o", tmp + "_scene.o", tmp + "_scene.hip"});
            objs.push_back(tmp + "_scene.o");
            jobs.emplace_back(spawn_to_log(cs, tmp + "_scene.log"), tmp + "_scene.log");
        }
        bool ok = true;
        std::string why;
        for (auto &j : jobs) {
            if (j.first < 0 || !wait_ok(j.first)) {
                ok = false;
                why += "\n--- " + j.second + "\n" + tail_of(j.second);
            }
        }
        if (ok) {
            std::vector<std::string> link = {hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp + ".so"};
            link.insert(link.end(), objs.begin(), objs.end());
            const pid_t lp = spawn_to_log(link, tmp + "_link.log");
            if (lp < 0 || !wait_ok(lp)) { ok = false; why += "\n--- link\n" + tail_of(tmp + "_link.log"); }
        }
        for (const char *sfx : {"_free.hip", "_free.o", "_scene.hip", "_scene.o"}) (void)unlink((tmp + sfx).c_str());
        if (!ok) {
            err = "building the kernels of tree [" + args + "] failed:" + why;
            std::lock_guard<std::mutex> lk(g_plugin_mu);
            g_jit_log = "failed: " + so;
            return nullptr;
        }
        for (const char *sfx : {"_free.log", "_scene.log", "_link.log"}) (void)unlink((tmp + sfx).c_str());
        if (rename((tmp + ".so").c_str(), so.c_str()) != 0) { err = "cannot move the built plugin into " + so; return nullptr; }   // (atomic: ranks may race)
        built = true;
    }
    if (!owned_and_private(so, false, why_not)) { err = "refusing the cached plugin: " + why_not; return nullptr; }
    void *h = dlopen(so.c_str(), RTLD_NOW | RTLD_LOCAL);
    auto entry = h ? (const TopoEntry *(*)())dlsym(h, "mppi_plugin_entry") : nullptr;
    auto sizes = h ? (void (*)(size_t *))dlsym(h, "mppi_plugin_sizes") : nullptr;
    size_t sz[4] = {0, 0, 0, 0};
    if (sizes) sizes(sz);
    if (!entry || sz[0] != sizeof(mppi_ctx) || sz[1] != sizeof(DevModel) || sz[2] != sizeof(TopoEntry) || sz[3] != sizeof(DevCfg)) {
        const std::string why = h ? "does not match this library" : std::string("cannot be loaded (") + dlerror() + ")";
        if (h) dlclose(h);
        if (!built && !retried) {   // a damaged or foreign file in the cache: thrown away and built afresh, once
            std::fprintf(stderr, "[mppi_hip] cached plugin %s %s: rebuilding\n", so.c_str(), why.c_str());
            (void)unlink(so.c_str());
            return jit_topology(nb, parents, with_scene, free_slots, err, true);
        }
        err = "the plugin " + so + " " + why;
        return nullptr;
    }
    const TopoEntry *e = entry();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    {
        std::lock_guard<std::mutex> lk(g_plugin_mu);
        g_plugins.push_back(e);
        char buf[64];
        std::snprintf(buf, sizeof buf, "%.1f", secs);
        g_jit_log = std::string(built ? "built " : "cached ") + so + " in " + buf + " s";
    }
    return e;
}

}  // namespace

// ------------------------------------------------------------------------------ C-ABI
extern "C" {

const char *mppi_last_error(void) { return g_err.c_str(); }
int mppi_abi_version(void) { return MPPI_ABI_VERSION; }
/* what the last on-demand build of a kinematic tree did ("built <plugin> in 34.1 s" / "cached ..." / "failed: ..."), or - called from
 * another thread while mppi_create blocks in one - what the running build is doing ("building <plugin> (tree [...]), 12 s so far");
 * empty: none so far */
int mppi_jit_info(char *buf, int buflen) {
    if (!buf || buflen < 1) return fail(MPPI_EINVAL, "null buffer");
    std::lock_guard<std::mutex> lk(g_plugin_mu);
    if (g_jit_building)
        std::snprintf(buf, buflen, "%s, %.0f s so far", g_jit_log.c_str(), std::chrono::duration<double>(std::chrono::steady_clock::now() - g_jit_t0).count());
    else
        std::snprintf(buf, buflen, "%s", g_jit_log.c_str());
    return MPPI_OK;
}
int mppi_device_count(int *count) {
    if (!count) return fail(MPPI_EINVAL, "null count");
    HIP_TRY(hipGetDeviceCount(count));
    return MPPI_OK;
}

int mppi_create(const mppi_model_t *model, const mppi_config_t *cfg, int device, mppi_ctx_t **out) {
    if (!model || !cfg || !out) return fail(MPPI_EINVAL, "null argument");
    std::string err;
    mppi_ctx *c = new mppi_ctx();
    c->model = *model;
    c->cfg = *cfg;
    c->device = device;
    if (!pack_model(*model, c->hm, err) || !pack_config(*cfg, c->hc, err)) {
        delete c;
        return fail(MPPI_EINVAL, err);
    }
    if (cfg->nu != model->nu) {
        delete c;
        return fail(MPPI_EINVAL, "config.nu != model.nu");
    }
    std::memset(&c->hk, 0, sizeof c->hk);
    int parents[MPPI_MAX_BODIES];
    for (int i = 0; i < c->hm.nb; i++) parents[i] = c->hm.b[i].k0.parent;
    c->topo = topology_string(c->hm.nb, parents);
    c->scene = is_scene(c->hm);
    hipError_t lds_err = hipSuccess;
    bool ok = false;
    // the launch-table row of this tree: shipped instantiations first, then plugins loaded earlier, then an on-demand build
    const TopoEntry *found = find_entry(c->hm.nb, parents, c->scene, c->hm.n_free);
    std::string jit_err;
    if (!found) found = jit_topology(c->hm.nb, parents, c->scene, c->hm.n_free > 2 ? kMaxFree : 2, jit_err);
    for (const TopoEntry *e = found; e != nullptr;) {
        ok = true;
        c->free_slots = e->free_slots > 0 ? e->free_slots : 2;
        c->launch_eval_cost = e->eval_cost;
        if (c->scene) {
            c->lds_bytes = sizeof(float) * kWave * (e->scene_lds_floats + 3 * (size_t)c->hm.n_rb + 5 * (size_t)c->hm.n_rnd);
            // (+ the records of the light bodies' pairs, mppi_scene.hpp scene_row_floats; the one-lane kernels keep them per lane)
            c->lds_bytes_quad = sizeof(float) * 16 * (e->scene_lds_floats + 3 * (size_t)c->hm.n_rb + 5 * (size_t)c->hm.n_rnd + 12 * (size_t)c->hm.n_shapes +
                                                      (c->hm.n_light_pairs != 0 ? (size_t)kLightFloats : 0));
            c->lds_bytes_table = sizeof(unsigned) * (size_t)scene_table_dwords(c->hm.n_shapes, c->hm.n_pairs);
            // rollouts: 4 lanes per sample (contact points dealt over the quad) unless MPPI_ROLLOUT=lane
            const char *mode = std::getenv("MPPI_ROLLOUT");
            c->quad = !(mode && std::string(mode) == "lane");
            // several moving-base robots in one env (ABI 7: one floating base per tree): the one-lane kernels carry them - the
            // shared-lane kernels keep ONE base (conf/mppi/multi-jackal.yaml asks for 100 samples: two wavefronts either way)
            if (c->hm.n_bases > 1) c->quad = false;
            // contact scenes: 8 lanes per sample (contact work dealt over an octet, K/8 wavefronts) unless MPPI_ROLLOUT=quad
            // (4 lanes per sample, the round-1 kernel) or =lane; one-sample contexts (the K = 1 world) keep the quad
            const bool oct = c->quad && !(mode && std::string(mode) == "quad") && cfg->num_samples >= 8;
            c->lanes_per_sample = !c->quad ? 1 : (oct ? 8 : 4);
            c->launch_rollout = c->quad ? (oct ? e->rollout_scene_oct : e->rollout_scene_quad) : e->rollout_scene;
            // short trees: the octet kernel with a helper wavefront per sample group (kSplitOctPair) unless MPPI_ROLLOUT=oct
            // (its dead-pair masks are two words, mppi_scene.hpp: a larger candidate list goes to the one-wavefront octet kernel)
            // (... and a scene with a light body's pairs - implicit on both bodies, mppi_scene.hpp "light bodies" - as well: the free
            // actors are solved AFTER the robot there, not next to it)
            c->helper_wave = oct && e->rollout_scene_pair != nullptr && !(mode && std::string(mode) == "oct") && c->hm.n_pairs <= kPairKernelMaxPairs &&
                             c->hm.n_light_pairs == 0;
            if (c->helper_wave) c->launch_rollout = e->rollout_scene_pair;
            if (oct) {  // (whole-horizon trajectories for host-side costs: the octet kernel)
                c->launch_rollout_traj = e->rollout_scene_traj;
                c->launch_materialise_traj = e->materialise_scene_traj;
            }
            c->launch_sim_step = c->quad ? e->sim_step_scene_quad : e->sim_step_scene;  // (the K = 1 world included: one quad)
            c->step_feeds_back = c->quad;
            c->launch_materialise = e->materialise_scene;
            // (large scenes - e.g. the 12-DoF mobile manipulator with table and block - do not fit the one-lane kernels' 64 rows
            // per wavefront into 160 KiB: those kernels are then simply not available, MPPI_ROLLOUT=lane is refused below)
            if (hipSetDevice(device) == hipSuccess) {
                c->lds_bytes_static = e->static_lds();
                if (c->lds_bytes_quad + c->lds_bytes_table + c->lds_bytes_static <= 160 * 1024)
                    lds_err = e->raise_lds(c->lds_bytes <= 160 * 1024 ? c->lds_bytes : 0, c->lds_bytes_quad + c->lds_bytes_table);
            }
        } else {
            // fixed-base contact-free scenes: one sample per 4-lane quad unless MPPI_ROLLOUT=lane asks for the
            // one-lane-per-sample kernel (kept for A/B measurements and as the reference arithmetic)
            // ... and, by default, with the articulated-body solve in the OCTET layout (mppi_oct.hpp: 8 lanes per sample, angular /
            // linear halves of every spatial quantity in two quads, K/8 wavefronts); MPPI_ROLLOUT=quad keeps 4 lanes per sample
            const char *mode = std::getenv("MPPI_ROLLOUT");
            c->quad = !(mode && std::string(mode) == "lane");
            // The octet kernel needs 257 registers: ONE wavefront per SIMD.  Up to K = 8 x SIMDs (8192 on the MI355X) that is all
            // it ever gets; beyond, its second wavefront per SIMD queues behind the first (measured, profiles/
            // r04h_wave_count_scan_oct.txt: 5.09 us per horizon step at K = 8192, 9.95 at 16384) while the 256-register quad
            // kernel co-hosts two (5.61 us at K = 16384): larger K keeps 4 lanes per sample.  MPPI_ROLLOUT=oct forces the octet.
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
            (void)hipGetLastError();
            const bool oct_fits = (cfg->num_samples + 7) / 8 <= 4 * cus || (mode && std::string(mode) == "oct");
            const bool oct = c->quad && !(mode && std::string(mode) == "quad") && cfg->num_samples >= 8 && oct_fits;
            c->lanes_per_sample = c->quad ? (oct ? 8 : 4) : 1;
            c->launch_rollout = c->quad ? (oct ? e->rollout_oct : e->rollout_quad) : e->rollout;
            c->launch_rollout_lane = e->rollout;  // cost programs on contact-free scenes run on the one-lane kernel
            if (c->quad) {
                c->launch_rollout_traj = oct ? e->rollout_oct_traj : e->rollout_traj;
                c->launch_materialise_traj = e->materialise_traj;
                c->launch_materialise_traj_link = e->materialise_traj_link;
            }
            c->launch_combine_world = e->combine_world;
            // many envs: quad step kernel; the K = 1 world (and tiny K) keeps the one-lane kernel
            // ... and, since round 5, the quad kernel again (one quad steps the world in ~4 us, the lone lane took ~12: the world loop of
            // the reference's examples waits for this kernel every control iteration); MPPI_WORLD_STEP=lane keeps the one-lane kernel
            const char *ws = std::getenv("MPPI_WORLD_STEP");
            const bool world_quad = cfg->num_samples == 1 && !(ws && std::string(ws) == "lane");
            c->launch_sim_step = (c->quad && (cfg->num_samples >= 64 || world_quad)) ? e->sim_step_quad : e->sim_step;
            c->launch_materialise = e->materialise;
        }
        break;
    }
    if (ok && ((c->quad ? 0 : c->lds_bytes) > 160 * 1024 || c->lds_bytes_quad + c->lds_bytes_table + c->lds_bytes_static > 160 * 1024)) {
        delete c;
        return fail(MPPI_EUNSUPPORTED, "contact scene needs more than 160 KiB of LDS per workgroup (dynamic rows + table + the kernels' static __shared__)");
    }
    if (ok && lds_err != hipSuccess) {
        std::string msg = std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(lds_err);
        delete c;
        return fail(MPPI_EHIP, msg);
    }
    if (!ok) {
        std::string t = c->topo;
        delete c;
        return fail(MPPI_EUNSUPPORTED, "kinematic tree " + t + " is not instantiated in this library and could not be built on demand: " + jit_err);
    }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) {
        delete c;
        return fail(MPPI_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    }
    {   // Fold of the wave records per XCD group inside the quad / octet rollout kernels (fold_group).  Measured on MI355X
        // (profiles/r02a_*, r02f_*): the agent-scope release every workgroup needs before its ticket costs the rollout +8 us.
        // Contact scenes (K/8 wavefronts = 1024 records of up to 272 floats for ONE combine workgroup: 32 us in the gripper
        // scene) win - 445 -> 460 Hz -, the 0.17-ms panda iteration (256 records, 10 us combine) loses 2.6 %: the fold is
        // on for contact scenes and off for the contact-free kernel; MPPI_FOLD=0 / =1 overrides either way.
        const char *f = std::getenv("MPPI_FOLD");
        c->fold = c->quad && (f ? std::string(f) == "1" : c->scene);
    }
    const int rc = create_buffers(c, cfg);
    if (rc != MPPI_OK) {  // (the error text is already set) nothing allocated so far may leak
        release_ctx(c);
        return rc;
    }
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.insert(c);
    }
    *out = c;
    return MPPI_OK;
}

int mppi_destroy(mppi_ctx_t *c) {
    if (!c) return MPPI_OK;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        if (g_live.erase(c) == 0) return fail(MPPI_EINVAL, "mppi_destroy: stale or foreign context handle");
    }
    release_ctx(c);
    return MPPI_OK;
}

int mppi_set_stream(mppi_ctx_t *c, void *hip_stream) {
    CTX_TRY(c);
    c->stream = (hipStream_t)hip_stream;
    return MPPI_OK;
}
int mppi_synchronize(mppi_ctx_t *c) {
    CTX_TRY(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MPPI_OK;
}

int mppi_set_state(mppi_ctx_t *c, const float *dof, const float *root) {
    CTX_TRY(c);
    // (ABI 8) the < 1 KB of state ride as kernel arguments of a one-wavefront launch, stream-ordered like the copies they
    // replace: no copy operation, no synchronise; the host buffers may be reused by the caller right away
    X0Args a;
    const int nd = dof ? 2 * c->n : 0, nr = root ? 13 * c->A : 0;
    if (nd) std::memcpy(a.v, dof, sizeof(float) * nd);
    if (nr) std::memcpy(a.v + 2 * MPPI_MAX_BODIES, root, sizeof(float) * nr);
    if (nd || nr) hipLaunchKernelGGL(k_set_x0, dim3(1), dim3(64), 0, c->stream, a, nd, nr, c->d_x0_dof, c->d_x0_root);
    return launch_check();
}
int mppi_set_state_dev(mppi_ctx_t *c, const float *dof, const float *root) {
    CTX_TRY(c);
    if (dof) HIP_TRY(hipMemcpyAsync(c->d_x0_dof, dof, sizeof(float) * 2 * c->n, hipMemcpyDeviceToDevice, c->stream));
    if (root) HIP_TRY(hipMemcpyAsync(c->d_x0_root, root, sizeof(float) * 13 * c->A, hipMemcpyDeviceToDevice, c->stream));
    return MPPI_OK;
}
int mppi_get_state(mppi_ctx_t *c, float *dof, float *root) {
    CTX_TRY(c);
    if (dof) HIP_TRY(hipMemcpyAsync(dof, c->d_x0_dof, sizeof(float) * 2 * c->n, hipMemcpyDeviceToHost, c->stream));
    if (root) HIP_TRY(hipMemcpyAsync(root, c->d_x0_root, sizeof(float) * 13 * c->A, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MPPI_OK;
}

int mppi_set_cost(mppi_ctx_t *c, const mppi_cost_t *cost) {
    CTX_TRY(c);
    if (!cost) return fail(MPPI_EINVAL, "null cost");
    std::string err;
    DevCost hk;
    if (!pack_cost(*cost, c->hm, hk, err)) return fail(cost->kind > MPPI_COST_PROGRAM || cost->kind < 0 || cost->kind == MPPI_COST_PROGRAM ? MPPI_EINVAL : MPPI_EUNSUPPORTED, err);
    c->hk = hk;
    // MPPI_COST_PROGRAM on a contact-free scene: the quad kernel's in-line costs stay as they are (its instruction stream is
    // the metric's); the interpreter lives in the one-lane kernel there, and in the scene kernels
    c->prog_lane = hk.kind == kCostProgram && !c->scene && c->quad && c->launch_rollout_lane != nullptr;
    HIP_TRY(hipMemcpyAsync(c->d_cost, &c->hk, sizeof(DevCost), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->has_cost = true;
    return MPPI_OK;
}

int mppi_sample(mppi_ctx_t *c, uint32_t index_base) {
    CTX_TRY(c);
    if (c->cfg.sampling != MPPI_SAMPLE_HALTON_SPLINE) return fail(MPPI_ESTATE, "mppi_sample: config.sampling is not halton-spline");
    hipLaunchKernelGGL(k_sample, dim3(c->n_waves), dim3(kWave), 0, c->stream, c->d_cfg, c->d_basis, c->d_sigma, c->cfg.n_knots, index_base, c->d_eps);
    c->eps_in = c->d_eps;
    return launch_check();
}
int mppi_sample_normal(mppi_ctx_t *c, uint32_t iteration) {
    CTX_TRY(c);
    if (c->cfg.sampling != MPPI_SAMPLE_NORMAL) return fail(MPPI_ESTATE, "mppi_sample_normal: config.sampling is not MPPI_SAMPLE_NORMAL");
    const int threads = c->K * c->nu;
    hipLaunchKernelGGL(k_sample_normal, dim3((threads + kWave - 1) / kWave), dim3(kWave), 0, c->stream, c->d_cfg, c->d_basis, c->d_sigma, c->cfg.n_knots,
                       (uint32_t)c->cfg.seed, iteration, c->d_eps);
    c->eps_in = c->d_eps;
    return launch_check();
}
int mppi_set_noise_dev(mppi_ctx_t *c, const float *eps_dev) {
    CTX_TRY(c);
    c->eps_in = eps_dev ? eps_dev : c->d_eps;
    return MPPI_OK;
}
int mppi_set_prior(mppi_ctx_t *c, const float *prior) {
    CTX_TRY(c);
    c->has_prior = prior != nullptr;
    if (prior) {
        HIP_TRY(hipMemcpyAsync(c->d_prior, prior, sizeof(float) * c->HN, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return MPPI_OK;
}
int mppi_set_prior_row(mppi_ctx_t *c, int t, const float *row) {
    CTX_TRY(c);
    if (!row || t < 0 || t >= c->H) return fail(MPPI_EINVAL, "mppi_set_prior_row: null row or t outside the horizon");
    c->has_prior = true;
    HIP_TRY(hipMemcpyAsync(c->d_prior + (size_t)t * c->nu, row, sizeof(float) * c->nu, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MPPI_OK;
}
int mppi_set_nominal(mppi_ctx_t *c, const float *U) {
    CTX_TRY(c);
    if (!U) return fail(MPPI_EINVAL, "null U");
    HIP_TRY(hipMemcpyAsync(c->d_U, U, sizeof(float) * c->HN, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MPPI_OK;
}
int mppi_set_filter(mppi_ctx_t *c, const float *F) {
    CTX_TRY(c);
    c->use_filter = F != nullptr;
    if (F) {
        HIP_TRY(hipMemcpyAsync(c->d_filter, F, sizeof(float) * c->H * c->H, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return MPPI_OK;
}
int mppi_set_lambda(mppi_ctx_t *c, double lambda) {
    CTX_TRY(c);
    if (!(lambda > 0) || !std::isfinite(lambda)) return fail(MPPI_EINVAL, "mppi_set_lambda: lambda must be positive and finite");
    c->cfg.lambda_ = lambda;
    c->hc.lambda = (float)lambda;
    c->hc.inv_lambda = (float)(1.0 / lambda);
    // (stream-ordered: the kernels of the iteration in flight keep the value they were launched with)
    HIP_TRY(hipMemcpyAsync(c->d_cfg, &c->hc, sizeof(DevCfg), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MPPI_OK;
}
int mppi_get_nominal(mppi_ctx_t *c, float *U) {
    CTX_TRY(c);
    HIP_TRY(hipMemcpyAsync(U, c->d_U, sizeof(float) * c->HN, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MPPI_OK;
}

int mppi_rollout(mppi_ctx_t *c) {
    CTX_TRY(c);
    if (!c->has_cost) return fail(MPPI_ESTATE, "mppi_rollout: no fused cost set (mppi_set_cost); use the mppi_sim_* path for host-side costs");
    {
        EvScope ev(c, 0);
        if (c->prog_lane) c->launch_rollout_lane(c);
        else c->launch_rollout(c);
    }
    c->partials_valid = true;
    if (c->prog_lane) {
        c->n_partials = c->n_waves;
        c->recs_cur = c->d_partials;
    } else if (c->fold) {  // the kernel's tail left one record per XCD group (ragged grids: one) in fold_out
        c->n_partials = c->n_quads % 16 == 0 ? kFoldGroups : 1;
        c->recs_cur = c->fold_out;
    } else {
        c->n_partials = c->quad ? c->n_quads : c->n_waves;
        c->recs_cur = c->d_partials;
    }
    return launch_check();
}
// ---- generic Objective mode, whole horizon at once -------------------------------------------------------------------
int mppi_rollout_trajectory(mppi_ctx_t *c) {
    CTX_TRY(c);
    if (!c->launch_rollout_traj)
        return fail(MPPI_EUNSUPPORTED, "mppi_rollout_trajectory: this context runs a kernel without the trajectory dump (MPPI_ROLLOUT=lane, or a contact scene with fewer than 8 samples); use the mppi_sim_* steps");
    const size_t HK = (size_t)c->H * c->K;
    if (!c->d_traj) {
        const size_t rows = c->scene ? 2 * (size_t)c->n + 13 + 13 * (size_t)c->free_slots + 3 * (size_t)c->hm.n_rb : 2 * (size_t)c->n;
        if (!c->d_cost_none) {
            ALLOC_TRY(c->d_cost_none, sizeof(DevCost));
            HIP_TRY(hipMemsetAsync(c->d_cost_none, 0, sizeof(DevCost), c->stream));  // kind = MPPI_COST_NONE
        }
        ALLOC_TRY(c->d_traj, sizeof(float) * rows * HK);  // (last: a failure here leaves nothing the next call would trust)
    }
    {
        EvScope ev(c, 0);
        c->launch_rollout_traj(c);
    }
    c->partials_valid = false;  // S holds the control cost only: the records follow the host-side costs (mppi_reduce)
    return launch_check();
}
int mppi_materialise_trajectory(mppi_ctx_t *c, float *dof, float *root, float *rb, float *cf) {
    CTX_TRY(c);
    if (!c->launch_materialise_traj || !c->d_traj) return fail(MPPI_ESTATE, "mppi_materialise_trajectory: no trajectory (mppi_rollout_trajectory)");
    c->launch_materialise_traj(c, dof, root, rb, cf);
    return launch_check();
}
/* (ABI 8) one rigid body of all H*K env-steps of the last mppi_rollout_trajectory as dense rows [H*K][13] - what
 * `sim.get_actor_link_by_name(actor, link)` of an Objective needs over the horizon, without the [H*K][n_rb][13] tensor around it.
 * MPPI_EUNSUPPORTED: contact scenes and rigid bodies that are no robot link (callers take the full tensor instead) */
int mppi_materialise_trajectory_link(mppi_ctx_t *c, int rb_index, float *out_dev) {
    CTX_TRY(c);
    if (!c->launch_materialise_traj || !c->d_traj) return fail(MPPI_ESTATE, "mppi_materialise_trajectory_link: no trajectory (mppi_rollout_trajectory)");
    if (!out_dev) return fail(MPPI_EINVAL, "null output");
    const int l = rb_index - c->hm.robot_first_rb;
    if (!c->launch_materialise_traj_link || c->scene || l < 0 || l >= c->hm.nl)
        return fail(MPPI_EUNSUPPORTED, "mppi_materialise_trajectory_link: a robot link of a contact-free scene");
    c->launch_materialise_traj_link(c, l, out_dev);
    return launch_check();
}
int mppi_reduce(mppi_ctx_t *c, float *record_out_dev) {
    CTX_TRY(c);
    if (!c->partials_valid) {  // generic mode: S came from host-side costs, build the per-wave records now
        EvScope ev(c, 1);
        const int n16 = (c->K + 15) / 16;
        if (n16 <= c->n_quads) {  // (the record buffer holds n_quads records: every context but the one-lane ones)
            hipLaunchKernelGGL(k_reduce_quad, dim3(n16), dim3(kWave), 0, c->stream, c->d_cfg, c->d_S, c->d_du, c->d_partials);
            c->n_partials = n16;
        } else {
            hipLaunchKernelGGL(k_reduce, dim3(c->n_waves), dim3(kWave), 0, c->stream, c->d_cfg, c->d_S, c->d_du, c->d_partials);
            c->n_partials = c->n_waves;
        }
        c->partials_valid = true;
        c->recs_cur = c->d_partials;
    }
    if (record_out_dev)  // ONE shard record (API of the first ABI; the fused path all-gathers its folded records instead, mppi_set_record_out)
        hipLaunchKernelGGL(k_combine, dim3(1), dim3(kCombineThreads), 0, c->stream, c->d_cfg, c->recs_cur, c->n_partials, 0, record_out_dev, c->d_U, c->d_action,
                           c->d_beta_eta, (const float *)nullptr);
    return launch_check();
}
/* (ABI 8) generic Objective mode with the whole horizon evaluated at once (mppi_rollout_trajectory + mppi_materialise_trajectory):
 * cost_dev [H][K] holds the stage costs of all env-steps (row t*K + k = env k after step t); S_k += sum_t gamma^t c_t[k] (+ the
 * control cost) and the per-wavefront records in ONE launch - instead of mppi_sim_accumulate_cost, mppi_sim_finish and mppi_reduce */
int mppi_reduce_horizon_costs(mppi_ctx_t *c, const float *cost_dev, float *record_out_dev) {
    CTX_TRY(c);
    if (!cost_dev) return fail(MPPI_EINVAL, "null cost");
    const int n16 = (c->K + 15) / 16;
    if (n16 > c->n_quads) return fail(MPPI_EUNSUPPORTED, "mppi_reduce_horizon_costs: this context keeps per-wavefront records of 64 samples (one-lane kernels); use mppi_sim_accumulate_cost + mppi_reduce");
    {
        EvScope ev(c, 1);
        hipLaunchKernelGGL(k_horizon_reduce_quad, dim3(n16), dim3(kWave), 0, c->stream, c->d_cfg, cost_dev, c->d_ctrl, c->d_S, c->d_du, c->d_partials);
    }
    c->n_partials = n16;
    c->partials_valid = true;
    c->recs_cur = c->d_partials;
    if (record_out_dev)
        hipLaunchKernelGGL(k_combine, dim3(1), dim3(kCombineThreads), 0, c->stream, c->d_cfg, c->recs_cur, c->n_partials, 0, record_out_dev, c->d_U, c->d_action,
                           c->d_beta_eta, (const float *)nullptr);
    return launch_check();
}
int mppi_shard_record_count(const mppi_ctx_t *c) {
    // (a cost program on a contact-free scene runs the one-lane kernel, which leaves per-wave records and folds nothing:
    // callers that asked before mppi_set_cost ask again after it)
    if (check_ctx(c) != MPPI_OK || !c->fold || c->prog_lane) return 0;
    return c->n_quads % 16 == 0 ? kFoldGroups : 1;
}
int mppi_set_record_out(mppi_ctx_t *c, float *records_dev) {
    CTX_TRY(c);
    if (!c->fold) return fail(MPPI_EUNSUPPORTED, "mppi_set_record_out: this context does not fold its wave records (lane kernels / MPPI_FOLD=0); use mppi_reduce");
    c->fold_out = records_dev ? records_dev : c->d_fold;
    return MPPI_OK;
}

// ---- direct exchange of the shard records (mailbox all-gather) ------------------------------------------------
// frees the mailbox buffers of a context and returns it to the "no mailbox" state (failed create, see below)
static void mailbox_release(mppi_ctx *c) {
    void *bufs[] = {c->d_inbox, c->d_peers, c->d_mb_seq, c->d_gathered, c->d_own_rec};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    c->d_inbox = nullptr; c->d_peers = nullptr; c->d_mb_seq = nullptr; c->d_gathered = nullptr; c->d_own_rec = nullptr; c->d_mb_status = nullptr;
    c->h_peers.clear();
    c->mb_rank = -1; c->mb_n = 0; c->mb_nrec = 0; c->inbox_bytes = 0; c->inbox_fine = false;
}
static int mailbox_alloc(mppi_ctx *c, int rank, int n_ranks) {
    const int cnt = mppi_shard_record_count(c);
    c->mb_rank = rank; c->mb_n = n_ranks; c->mb_nrec = cnt > 0 ? cnt : 1;
    c->inbox_bytes = MailboxHeader::bytes(n_ranks, c->mb_nrec, c->RF);
    HIP_TRY(hipSetDevice(c->device));
    // fine-grained device memory: stores of a peer GPU become visible while this GPU's kernels run (coarse-grained memory is
    // only guaranteed at kernel boundaries); plain device memory when the runtime refuses - good for the contexts of ONE device
    // (tests), never handed to another process: mppi_mailbox_ipc_handle refuses a coarse-grained inbox
    c->inbox_fine = hipExtMallocWithFlags(&c->d_inbox, c->inbox_bytes, hipDeviceMallocFinegrained) == hipSuccess;
    if (!c->inbox_fine) {
        (void)hipGetLastError();
        c->d_inbox = nullptr;
        HIP_TRY(hipMalloc(&c->d_inbox, c->inbox_bytes));
    }
    HIP_TRY(hipMemset(c->d_inbox, 0, c->inbox_bytes));
    ALLOC_TRY(c->d_peers, sizeof(void *) * n_ranks);
    ALLOC_TRY(c->d_mb_seq, sizeof(unsigned));
    ALLOC_TRY(c->d_gathered, sizeof(float) * (size_t)n_ranks * c->mb_nrec * c->RF);
    // this shard's records when the rollout's tail has not folded them (generic Objective mode, ragged grids, one-lane
    // kernels): record 0 = the reduced shard record, records 1.. stay neutral (eta = 0: the combine skips them)
    ALLOC_TRY(c->d_own_rec, sizeof(float) * (size_t)c->mb_nrec * c->RF);
    // status word (1: a wait timed out) in the mapped host block next to the action and its sequence number: the host reads it
    // without a copy operation
    c->h_action[17] = 0.f;
    c->d_mb_status = reinterpret_cast<unsigned *>(c->hc.action_mirror + 17);
    c->h_peers.assign(n_ranks, nullptr);
    c->h_peers[rank] = c->d_inbox;
    c->peers_dirty = true;
    return MPPI_OK;
}
int mppi_mailbox_create(mppi_ctx_t *c, int rank, int n_ranks) {
    CTX_TRY(c);
    if (n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks) return fail(MPPI_EINVAL, "mppi_mailbox_create: need 0 <= rank < n_ranks <= 64");
    if (c->d_inbox) return fail(MPPI_ESTATE, "mppi_mailbox_create: this context already has a mailbox");
    const int rc = mailbox_alloc(c, rank, n_ranks);
    if (rc != MPPI_OK) mailbox_release(c);  // failure-atomic: a retry starts from scratch, nothing half-built is ever launched on
    return rc;
}
int mppi_mailbox_info(mppi_ctx_t *c, int *fine_grained, int *records_per_rank, int *n_ranks) {
    CTX_TRY(c);
    if (!c->d_inbox) return fail(MPPI_ESTATE, "no mailbox (mppi_mailbox_create)");
    if (fine_grained) *fine_grained = c->inbox_fine ? 1 : 0;
    if (records_per_rank) *records_per_rank = c->mb_nrec;
    if (n_ranks) *n_ranks = c->mb_n;
    return MPPI_OK;
}
int mppi_mailbox_ptr(mppi_ctx_t *c, void **inbox_dev, size_t *bytes) {
    CTX_TRY(c);
    if (!c->d_inbox) return fail(MPPI_ESTATE, "no mailbox (mppi_mailbox_create)");
    if (inbox_dev) *inbox_dev = c->d_inbox;
    if (bytes) *bytes = c->inbox_bytes;
    return MPPI_OK;
}
int mppi_mailbox_ipc_handle(mppi_ctx_t *c, void *handle64) {
    CTX_TRY(c);
    if (!c->d_inbox || !handle64) return fail(MPPI_ESTATE, "no mailbox (mppi_mailbox_create)");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    // peers of another process / GPU store into the inbox while this GPU's kernels poll it: that needs fine-grained memory
    // (MPPI_MAILBOX_ALLOW_COARSE=1: processes that share ONE device, where a coarse-grained allocation is coherent in its L2)
    const char *coarse_ok = std::getenv("MPPI_MAILBOX_ALLOW_COARSE");
    if (!c->inbox_fine && !(coarse_ok && std::string(coarse_ok) == "1"))
        return fail(MPPI_EUNSUPPORTED, "mppi_mailbox_ipc_handle: the inbox is not fine-grained device memory (hipExtMallocWithFlags refused); use the RCCL all-gather");
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, c->d_inbox));
    std::memcpy(handle64, &h, 64);
    return MPPI_OK;
}
int mppi_mailbox_set_peer(mppi_ctx_t *c, int peer_rank, void *inbox_dev) {
    CTX_TRY(c);
    if (!c->d_inbox || peer_rank < 0 || peer_rank >= c->mb_n || !inbox_dev) return fail(MPPI_EINVAL, "mppi_mailbox_set_peer: bad rank / pointer, or no mailbox");
    c->h_peers[peer_rank] = inbox_dev;
    c->peers_dirty = true;
    return MPPI_OK;
}
int mppi_mailbox_open(mppi_ctx_t *c, int peer_rank, const void *handle64) {
    CTX_TRY(c);
    if (!c->d_inbox || peer_rank < 0 || peer_rank >= c->mb_n || !handle64) return fail(MPPI_EINVAL, "mppi_mailbox_open: bad rank / handle, or no mailbox");
    if (peer_rank == c->mb_rank) return MPPI_OK;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle64, 64);
    void *p = nullptr;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    c->ipc_opened.push_back(p);
    // pre-flight: the mapping must take a write and give it back BEFORE any kernel stores through it (a kernel's store to an
    // unmapped peer address is a memory fault that ends the process; a failed copy is an error code).  Word 1 of this rank's flag
    // line in the peer's inbox is padding no kernel touches.
    {
        unsigned *probe_at = reinterpret_cast<unsigned *>(p) + (size_t)kMailboxFlagStride * c->mb_rank + 1;
        const unsigned probe = 0x4d500000u + (unsigned)c->mb_rank;
        unsigned back = 0;
        HIP_TRY(hipMemcpy(probe_at, &probe, sizeof probe, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(&back, probe_at, sizeof back, hipMemcpyDeviceToHost));
        if (back != probe) return fail(MPPI_EHIP, "mppi_mailbox_open: the peer's inbox does not hold what was written to it (no coherent mapping)");
    }
    c->h_peers[peer_rank] = p;
    c->peers_dirty = true;
    return MPPI_OK;
}
int mppi_mailbox_gathered(mppi_ctx_t *c, float **records_dev, int *n_records) {
    CTX_TRY(c);
    if (!c->d_inbox) return fail(MPPI_ESTATE, "no mailbox (mppi_mailbox_create)");
    if (records_dev) *records_dev = c->d_gathered;
    if (n_records) *n_records = c->mb_n * c->mb_nrec;
    return MPPI_OK;
}
/* publish this shard's records of the current rollout into every rank's inbox, wait for all ranks' records of this
 * iteration: afterwards mppi_update(ctx, gathered, n) / mppi_update_step_world combine them (mppi_mailbox_gathered) */
static int mailbox_own_records(mppi_ctx_t *c, const float **own_out) {
    if (!c->d_inbox) return fail(MPPI_ESTATE, "no mailbox (mppi_mailbox_create)");
    for (void *p : c->h_peers)
        if (!p) return fail(MPPI_ESTATE, "mppi_exchange: a peer's inbox is not connected (mppi_mailbox_set_peer / mppi_mailbox_open)");
    if (c->peers_dirty) {
        HIP_TRY(hipMemcpyAsync(c->d_peers, c->h_peers.data(), sizeof(void *) * c->mb_n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->peers_dirty = false;
    }
    const float *own;
    const int folded = mppi_shard_record_count(c);
    if (c->partials_valid && folded > 0 && folded == c->mb_nrec) {
        own = c->fold_out;  // the rollout's tail has folded the wave records already
    } else {
        // nothing folded (generic Objective mode: S came from host-side costs; a ragged grid; the one-lane kernel of a cost
        // program) although the mailbox may have been sized for folded records: ONE reduced shard record in slot 0, the other
        // slots keep eta = 0 and drop out of the combine
        int rc = mppi_reduce(c, c->d_own_rec);
        if (rc) return rc;
        own = c->d_own_rec;
    }
    *own_out = own;
    return MPPI_OK;
}
int mppi_exchange_publish(mppi_ctx_t *c) {
    CTX_TRY(c);
    const float *own = nullptr;
    int rc = mailbox_own_records(c, &own);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mailbox_publish, dim3(1), dim3(256), 0, c->stream, own, c->mb_nrec, c->RF, c->mb_rank, c->mb_n, (void *const *)c->d_peers, c->d_mb_seq);
    return launch_check();
}
int mppi_exchange_wait(mppi_ctx_t *c) {
    CTX_TRY(c);
    if (!c->d_inbox) return fail(MPPI_ESTATE, "no mailbox (mppi_mailbox_create)");
    // a peer that never publishes ends the wait after ~2 s of the 100 MHz wall clock instead of hanging the device
    hipLaunchKernelGGL(k_mailbox_wait, dim3(1), dim3(256), 0, c->stream, c->d_inbox, c->mb_nrec, c->RF, c->mb_n, (const unsigned *)c->d_mb_seq, c->d_gathered,
                       c->d_mb_status, 200000000ull);
    return launch_check();
}
int mppi_exchange(mppi_ctx_t *c) {
    CTX_TRY(c);
    // per-wavefront records of a fused rollout and a mailbox sized for ONE shard record: reduce and exchange in one launch
    if (c->d_inbox && c->partials_valid && c->mb_nrec == 1 && !(mppi_shard_record_count(c) > 0 && mppi_shard_record_count(c) == c->mb_nrec)) {
        bool connected = true;
        for (void *p : c->h_peers) connected = connected && p != nullptr;
        if (connected && !c->peers_dirty) {
            hipLaunchKernelGGL(k_reduce_exchange, dim3(1), dim3(kCombineThreads), 0, c->stream, c->d_cfg, (const float *)c->recs_cur, c->n_partials, c->d_own_rec, c->RF,
                               c->mb_rank, c->mb_n, (void *const *)c->d_peers, c->d_mb_seq, c->d_inbox, c->d_gathered, c->d_mb_status, 200000000ull);
            return launch_check();
        }
    }
    const float *own = nullptr;
    int rc = mailbox_own_records(c, &own);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mailbox_exchange, dim3(1), dim3(256), 0, c->stream, own, c->mb_nrec, c->RF, c->mb_rank, c->mb_n, (void *const *)c->d_peers, c->d_mb_seq,
                       c->d_inbox, c->d_gathered, c->d_mb_status, 200000000ull);
    return launch_check();
}
/* mppi_exchange + mppi_update_step_world; ONE launch where the rollout left per-wavefront records, the mailbox carries one
 * shard record per rank and the fused combine + world kernel exists (fixed-base contact-free scenes) */
int mppi_exchange_update_step_world(mppi_ctx_t *c, mppi_ctx_t *world) {
    CTX_TRY(c);
    CTX_TRY(world);
    bool fused = c->d_inbox && c->partials_valid && c->mb_nrec == 1 && !c->peers_dirty && c->launch_combine_world != nullptr && !c->scene && !world->scene &&
                 world->K == 1 && world->n == c->n && world->A == c->A && world->device == c->device &&
                 !(mppi_shard_record_count(c) > 0 && mppi_shard_record_count(c) == c->mb_nrec);
    for (void *p : c->h_peers) fused = fused && p != nullptr;
    if (!fused) {
        int rc = mppi_exchange(c);
        if (rc) return rc;
        return mppi_update_step_world(c, c->d_gathered, c->mb_n * c->mb_nrec, world);
    }
    {
        EvScope ev(c, 2);
        c->launch_combine_world(c, nullptr, 0, world);
    }
    c->seq_expected++;
    return launch_check();
}
int mppi_exchange_status(mppi_ctx_t *c, int *timed_out) {
    CTX_TRY(c);
    if (!c->d_inbox || !timed_out) return fail(MPPI_ESTATE, "no mailbox (mppi_mailbox_create)");
    HIP_TRY(hipStreamSynchronize(c->stream));  // (the waits enqueued so far have run; the word itself is mapped host memory)
    std::atomic_thread_fence(std::memory_order_acquire);
    volatile unsigned *word = reinterpret_cast<volatile unsigned *>(c->h_action + 17);
    *timed_out = (int)*word;
    // read-and-clear: the word reports the waits since the LAST call - one transient stall (a first-iteration graph capture, a
    // debugger pause on a peer) must not mark every later iteration late; the stream is idle here, nobody else writes the word
    *word = 0u;
    std::atomic_thread_fence(std::memory_order_release);
    return MPPI_OK;
}

int mppi_note_graph_update(mppi_ctx_t *c, int n) {
    CTX_TRY(c);
    c->seq_expected += (unsigned)n;
    return MPPI_OK;
}
int mppi_record_floats(const mppi_ctx_t *c) { return check_ctx(c) == MPPI_OK ? c->RF : 0; }
int mppi_record_dev(mppi_ctx_t *c, float **record_dev) {
    CTX_TRY(c);
    hipLaunchKernelGGL(k_combine, dim3(1), dim3(kCombineThreads), 0, c->stream, c->d_cfg, c->recs_cur ? c->recs_cur : c->d_partials, c->n_partials, 0, c->d_record, c->d_U, c->d_action, c->d_beta_eta, (const float *)nullptr);
    *record_dev = c->d_record;
    return launch_check();
}
int mppi_update(mppi_ctx_t *c, const float *records_dev, int n_records) {
    CTX_TRY(c);
    const float *recs = records_dev ? records_dev : (c->recs_cur ? c->recs_cur : c->d_partials);
    int n = records_dev ? n_records : c->n_partials;
    if (n < 1) return fail(MPPI_EINVAL, "n_records < 1");
    {
        EvScope ev(c, 2);
        hipLaunchKernelGGL(k_combine, dim3(1), dim3(kCombineThreads), 0, c->stream, c->d_cfg, recs, n, 1, (float *)nullptr, c->d_U, c->d_action, c->d_beta_eta,
                           c->use_filter ? (const float *)c->d_filter : (const float *)nullptr);
    }
    c->seq_expected++;  // the kernel publishes this number next to the action (mppi_wait_action)
    return launch_check();
}
/* closed-loop tail: mppi_update + mppi_world_step_from + mppi_set_state_from_world, one launch where possible */
int mppi_update_step_world(mppi_ctx_t *c, const float *records_dev, int n_records, mppi_ctx_t *world) {
    CTX_TRY(c);
    CTX_TRY(world);
    if (world->K != 1 || world->n != c->n || world->A != c->A || world->device != c->device)
        return fail(MPPI_EINVAL, "world must be a K=1 context of the same scene on the same device");
    if (c->launch_combine_world == nullptr || c->scene || world->scene) {  // contact scenes
        int rc;
        if ((rc = mppi_update(c, records_dev, n_records))) return rc;
        if (world->scene && world->step_feeds_back && world->K == 1) {
            // TWO launches: combine + update, then the world's step kernel, which writes the planner's next start state itself
            world->fb_dof = c->d_x0_dof;
            world->fb_root = c->d_x0_root;
            world->launch_sim_step(world, 1, 0, c->d_action);
            world->fb_dof = world->fb_root = nullptr;
            return launch_check();
        }
        if ((rc = mppi_world_step_from(world, c))) return rc;
        return mppi_set_state_from_world(c, world);
    }
    const float *recs = records_dev ? records_dev : (c->recs_cur ? c->recs_cur : c->d_partials);
    const int n = records_dev ? n_records : c->n_partials;
    if (n < 1) return fail(MPPI_EINVAL, "n_records < 1");
    {
        EvScope ev(c, 2);
        c->launch_combine_world(c, recs, n, world);
    }
    c->seq_expected++;
    return launch_check();
}
int mppi_wait_action(mppi_ctx_t *c, float *action) {
    CTX_TRY(c);
    if (!action) return fail(MPPI_EINVAL, "null action");
    const volatile unsigned *seq = reinterpret_cast<const volatile unsigned *>(c->h_action + 16);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; *seq != c->seq_expected; spins++) {
        if ((spins & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) {
            hipError_t e = hipStreamSynchronize(c->stream);  // a kernel fault surfaces here
            if (e != hipSuccess) return fail(MPPI_EHIP, std::string("mppi_wait_action: ") + hipGetErrorString(e));
            if (*seq != c->seq_expected) return fail(MPPI_ESTATE, "mppi_wait_action: the update kernel never published its action");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    std::memcpy(action, c->h_action, sizeof(float) * c->nu);
    return MPPI_OK;
}
int mppi_get_action(mppi_ctx_t *c, float *action) {
    CTX_TRY(c);
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::memcpy(action, c->h_action, sizeof(float) * c->nu);
    return MPPI_OK;
}
int mppi_action_dev(mppi_ctx_t *c, float **action_dev) {
    CTX_TRY(c);
    *action_dev = c->d_action;
    return MPPI_OK;
}
int mppi_command(mppi_ctx_t *c, float *action) {
    int rc;
    if ((rc = mppi_rollout(c))) return rc;
    if ((rc = mppi_reduce(c, nullptr))) return rc;
    if ((rc = mppi_update(c, nullptr, 1))) return rc;
    return action ? mppi_get_action(c, action) : MPPI_OK;
}

static int d2h(mppi_ctx_t *c, float *dst, const float *src, size_t nfloats) {
    CTX_TRY(c);
    if (!dst) return fail(MPPI_EINVAL, "null destination");
    HIP_TRY(hipMemcpyAsync(dst, src, sizeof(float) * nfloats, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MPPI_OK;
}
int mppi_get_costs(mppi_ctx_t *c, float *S) { return d2h(c, S, c ? c->d_S : nullptr, c ? c->K : 0); }
int mppi_get_weights_stats(mppi_ctx_t *c, float *be) { return d2h(c, be, c ? c->d_beta_eta : nullptr, 2); }
int mppi_get_rollouts(mppi_ctx_t *c, float *viz) {
    CTX_TRY(c);  // (a stale handle is reported, not dereferenced)
    if (!c->cfg.want_rollouts) return fail(MPPI_ESTATE, "config.want_rollouts is off");
    // device layout is sample-minor [H][3][K] (full-line coalesced stores); hand out [H][K][3]
    std::vector<float> tmp((size_t)c->H * 3 * c->K);
    int rc = d2h(c, tmp.data(), c->d_viz, tmp.size());
    if (rc) return rc;
    for (int t = 0; t < c->H; t++)
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < c->K; k++) viz[((size_t)t * c->K + k) * 3 + j] = tmp[((size_t)t * 3 + j) * c->K + k];
    return MPPI_OK;
}
int mppi_get_perturbations(mppi_ctx_t *c, float *du) { return d2h(c, du, c ? c->d_du : nullptr, c ? (size_t)c->HN * c->K : 0); }
int mppi_get_noise(mppi_ctx_t *c, float *eps) { return d2h(c, eps, c ? c->eps_in : nullptr, c ? (size_t)c->HN * c->K : 0); }

/* parity / debug: the context's cost program on caller-given simulator answers (host arrays, reference layouts) */
int mppi_eval_cost(mppi_ctx_t *c, int n, const float *dof, const float *root, const float *rb, const float *cf, float *cost_out) {
    CTX_TRY(c);
    if (n < 1 || !dof || !root || !rb || !cf || !cost_out) return fail(MPPI_EINVAL, "mppi_eval_cost: null argument or n < 1");
    if (!c->has_cost || c->hk.kind != kCostProgram) return fail(MPPI_ESTATE, "mppi_eval_cost: the context's cost is not a MPPI_COST_PROGRAM (mppi_set_cost)");
    if (!c->launch_eval_cost) return fail(MPPI_EUNSUPPORTED, "mppi_eval_cost: not available for this kinematic tree");
    const size_t sz[5] = {(size_t)n * 2 * c->n, (size_t)n * 13 * c->A, (size_t)n * 13 * c->B, (size_t)n * 3 * c->B, (size_t)n};
    const float *src[4] = {dof, root, rb, cf};
    float *d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int rc = MPPI_OK;
    for (int i = 0; i < 5 && rc == MPPI_OK; i++) rc = dev_alloc(&d[i], sizeof(float) * sz[i]);
    for (int i = 0; i < 4 && rc == MPPI_OK; i++)
        if (hipMemcpyAsync(d[i], src[i], sizeof(float) * sz[i], hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = fail(MPPI_EHIP, "mppi_eval_cost: upload failed");
    if (rc == MPPI_OK) {
        c->launch_eval_cost(c, n, d[0], d[1], d[2], d[3], d[4]);
        rc = launch_check();
    }
    if (rc == MPPI_OK && (hipMemcpyAsync(cost_out, d[4], sizeof(float) * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess))
        rc = fail(MPPI_EHIP, "mppi_eval_cost: download failed");
    for (float *p : d)
        if (p) (void)hipFree(p);
    return rc;
}

int mppi_sim_reset(mppi_ctx_t *c) {
    CTX_TRY(c);
    c->partials_valid = false;
    hipLaunchKernelGGL(k_sim_reset, dim3((c->K + 255) / 256), dim3(256), 0, c->stream, c->K, c->n, c->d_x0_dof, c->d_q, c->d_qd, c->d_S, c->d_ctrl);
    if (c->scene)
        hipLaunchKernelGGL(k_sim_reset_scene, dim3((c->K + 255) / 256), dim3(256), 0, c->stream, c->d_model, c->K, c->d_x0_root, c->d_base, c->d_fr, c->d_cf, c->free_slots);
    return launch_check();
}
int mppi_sim_step(mppi_ctx_t *c, const float *u_dev, int u_is_shared) {
    CTX_TRY(c);
    if (!u_dev) return fail(MPPI_EINVAL, "null command");
    c->launch_sim_step(c, u_is_shared ? 1 : 0, 0, u_dev);
    return launch_check();
}
/* (ABI 8) one command for every env, handed over by HOST pointer: it is written into a ring slot of the context's mapped host
 * block and the step kernel reads it through the mapped pointer - what apply_robot_cmd + step of a K = 1 world need per control
 * iteration (reference examples/<x>/world.py:42-44), without a host-to-device copy operation */
int mppi_sim_step_host(mppi_ctx_t *c, const float *u_host) {
    CTX_TRY(c);
    if (!u_host) return fail(MPPI_EINVAL, "null command");
    const unsigned slot = c->io_cmd_next++ % kIoCmdSlots;
    // back-pressure: a slot is written again only after the kernels of the previous lap have read theirs - a caller that steps without
    // ever reading state (many envs, contact scenes with multiplied substeps) could otherwise get more than kIoCmdSlots launches
    // ahead of the stream, and steps would silently run with later commands.  One wait per lap; the loops that read the state
    // every iteration (the reference's world loop) find the stream idle
    if (slot == 0 && c->io_cmd_next > 1) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(MPPI_EHIP, std::string("mppi_sim_step_host: ") + hipGetErrorString(e));
    }
    std::memcpy(c->h_io + kIoCmd + 16 * slot, u_host, sizeof(float) * c->nu);
    std::atomic_thread_fence(std::memory_order_release);
    c->launch_sim_step(c, 1, 0, c->d_io + kIoCmd + 16 * slot);
    return launch_check();
}
/* (ABI 8) K = 1 world: mppi_sim_materialise whose kernel ALSO mirrors env 0's dof [2n] and root [A][13] rows into the context's
 * mapped host block and publishes a sequence number behind them; mppi_mirror_wait polls the number and copies them out - the
 * `torch_to_bytes(sim._dof_state)` of the reference's world loop (examples/<x>/world.py:35-39) without a device-to-host copy
 * operation or a stream synchronise */
int mppi_sim_materialise_mirror(mppi_ctx_t *c, float *dof, float *root, float *rb, float *cf) {
    CTX_TRY(c);
    if (c->K != 1 || !dof || !root) return fail(MPPI_EINVAL, "mppi_sim_materialise_mirror: a K = 1 context and its dof / root tensors");
    c->io_mirror_seq++;
    c->mirror_armed = true;
    c->launch_materialise(c, dof, root, rb, cf);
    c->mirror_armed = false;
    return launch_check();
}
int mppi_mirror_wait(mppi_ctx_t *c, float *dof_host, float *root_host) {
    CTX_TRY(c);
    if (c->io_mirror_seq == 0) return fail(MPPI_ESTATE, "mppi_mirror_wait: nothing mirrored (mppi_sim_materialise_mirror)");
    const volatile unsigned *seq = reinterpret_cast<const volatile unsigned *>(c->h_io);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0; *seq != c->io_mirror_seq; spins++) {
        if ((spins & 0xFFFF) == 0xFFFF && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) {
            hipError_t e = hipStreamSynchronize(c->stream);  // a kernel fault surfaces here
            if (e != hipSuccess) return fail(MPPI_EHIP, std::string("mppi_mirror_wait: ") + hipGetErrorString(e));
            if (*seq != c->io_mirror_seq) return fail(MPPI_ESTATE, "mppi_mirror_wait: the mirror kernel never published");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (dof_host) std::memcpy(dof_host, c->h_io + kIoDof, sizeof(float) * 2 * c->n);
    if (root_host) std::memcpy(root_host, c->h_io + kIoDof + kIoDofFloats, sizeof(float) * 13 * c->A);
    return MPPI_OK;
}
int mppi_sim_step_horizon(mppi_ctx_t *c, int t) {
    CTX_TRY(c);
    if (t < 0 || t >= c->H) return fail(MPPI_EINVAL, "t outside the horizon");
    c->launch_sim_step(c, 2, t, nullptr);
    return launch_check();
}
int mppi_sim_materialise(mppi_ctx_t *c, float *dof, float *root, float *rb, float *cf) {
    CTX_TRY(c);
    c->launch_materialise(c, dof, root, rb, cf);
    return launch_check();
}
int mppi_sim_accumulate_cost(mppi_ctx_t *c, int t, const float *cost_dev) {
    CTX_TRY(c);
    if (!cost_dev) return fail(MPPI_EINVAL, "null cost");
    c->partials_valid = false;
    float disc = std::pow((float)c->cfg.rollout_var_discount, (float)t);
    hipLaunchKernelGGL(k_accumulate_cost, dim3((c->K + 255) / 256), dim3(256), 0, c->stream, c->K, disc, cost_dev, c->d_S);
    return launch_check();
}
int mppi_sim_finish(mppi_ctx_t *c) {
    CTX_TRY(c);
    hipLaunchKernelGGL(k_sim_finish, dim3((c->K + 255) / 256), dim3(256), 0, c->stream, c->K, c->d_ctrl, c->d_S);
    return launch_check();
}
int mppi_world_step_from(mppi_ctx_t *world, mppi_ctx_t *planner) {
    CTX_TRY(world);
    CTX_TRY(planner);
    if (world->nu != planner->nu || world->device != planner->device) return fail(MPPI_EINVAL, "world/planner mismatch");
    world->launch_sim_step(world, 1, 0, planner->d_action);
    return launch_check();
}
int mppi_set_state_from_world(mppi_ctx_t *planner, mppi_ctx_t *world) {
    CTX_TRY(planner);
    CTX_TRY(world);
    if (world->K != 1 || world->n != planner->n || world->A != planner->A) return fail(MPPI_EINVAL, "world must be a K=1 context of the same scene");
    hipLaunchKernelGGL(k_state_from_world, dim3(1), dim3(64), 0, planner->stream, planner->n, world->d_q, world->d_qd, planner->d_x0_dof);
    HIP_TRY(hipMemcpyAsync(planner->d_x0_root, world->d_x0_root, sizeof(float) * 13 * planner->A, hipMemcpyDeviceToDevice, planner->stream));
    if (world->scene)  // the world's robot base and free actors have moved: K = 1, so sample-minor rows are plain arrays
        hipLaunchKernelGGL(k_root_from_world, dim3(1), dim3(64), 0, planner->stream, planner->d_model, world->d_base, world->d_fr, planner->d_x0_root);
    return launch_check();
}

int mppi_set_wave_clock(mppi_ctx_t *c, int on) {
    CTX_TRY(c);
    // (the instrumented build keeps 12 section counters per wavefront behind the [start, end] rows)
    if (on && !c->d_wave_clk) ALLOC_TRY(c->d_wave_clk, sizeof(unsigned long long) * (2 + 16) * (size_t)(c->n_quads > c->n_waves ? c->n_quads : c->n_waves));
    c->wave_clk_on = on != 0;
    return MPPI_OK;
}
int mppi_get_wave_clock(mppi_ctx_t *c, uint64_t *start_end_host, int n_wavefronts) {
    CTX_TRY(c);
    if (!c->d_wave_clk || !start_end_host) return fail(MPPI_ESTATE, "mppi_get_wave_clock: not enabled (mppi_set_wave_clock)");
    const int n = c->quad ? c->n_quads : c->n_waves;
    if (n_wavefronts != n) return fail(MPPI_EINVAL, "mppi_get_wave_clock: the rollout kernel runs " + std::to_string(n) + " wavefronts");
    HIP_TRY(hipMemcpyAsync(start_end_host, c->d_wave_clk, sizeof(unsigned long long) * 2 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MPPI_OK;
}
#if defined(MPPI_SECTION_CLOCKS)
/* instrumented build only: [n_wavefronts][12] shader-clock ticks per section of the last scene rollout */
int mppi_get_section_clock(mppi_ctx_t *c, uint64_t *host, int n_wavefronts) {
    CTX_TRY(c);
    if (!c->d_wave_clk || !host || n_wavefronts != c->n_quads) return fail(MPPI_ESTATE, "mppi_get_section_clock: enable mppi_set_wave_clock first");
    HIP_TRY(hipMemcpyAsync(host, c->d_wave_clk + 2 * (size_t)c->n_quads, sizeof(unsigned long long) * 16 * (size_t)c->n_quads, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MPPI_OK;
}
#endif
int mppi_set_profiling(mppi_ctx_t *c, int on) {
    CTX_TRY(c);
    c->profiling = on != 0;
    c->profile_period = on > 1 ? on : 1;
    for (int w = 0; w < 3; w++) c->ev_used[w] = c->ev_seen[w] = 0;
    // the event pairs of the first brackets are created HERE, not inside the profiled region (EvScope grows the pool on demand
    // beyond them): two hipEventCreate per bracket and kernel sat between the launches of a closed loop that waits for every action
    if (on)
        for (int w = 0; w < 3; w++)
            while (c->ev[w].size() < 512) {
                hipEvent_t a, b;
                if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { (void)hipGetLastError(); break; }
                c->ev[w].emplace_back(a, b);
            }
    return MPPI_OK;
}
/* average hipEvent duration (ms) of the launches of kernel `which` since mppi_set_profiling(ctx,1) */
int mppi_kernel_ms(mppi_ctx_t *c, int which, float *ms) {
    CTX_TRY(c);
    if (which < 0 || which > 2 || !ms) return fail(MPPI_EINVAL, "bad argument");
    HIP_TRY(hipStreamSynchronize(c->stream));
    size_t n = c->ev_used[which];
    if (n == 0) return fail(MPPI_ESTATE, "no profiled launches recorded");
    double tot = 0;
    for (size_t i = 0; i < n; i++) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, c->ev[which][i].first, c->ev[which][i].second));
        tot += t;
    }
    *ms = (float)(tot / n);
    c->ev_used[which] = 0;
    return MPPI_OK;
}
int mppi_kernel_info(mppi_ctx_t *c, char *buf, int buflen) {
    CTX_TRY(c);
    std::snprintf(buf, buflen, "topology=%s rollout=%s K=%d H=%d nu=%d waves=%d block=%d bytes_alg=%zu", c->topo.c_str(), c->scene ? (c->helper_wave ? "scene-oct-pair" : (c->lanes_per_sample == 8 ? "scene-oct" : (c->quad ? "scene-quad" : "scene"))) : (c->quad ? (c->lanes_per_sample == 8 ? "oct" : "quad") : "lane"), c->K, c->H, c->nu, c->quad ? c->n_quads * ((c->lanes_per_sample == 8 && !c->scene) ? 2 : 1) : c->n_waves, kWave,
                  (size_t)4 * (3 * (size_t)c->K * c->HN + 2 * (size_t)c->K + c->HN));
    return MPPI_OK;
}

}  // extern "C"
