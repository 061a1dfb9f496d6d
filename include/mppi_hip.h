/*
 * mppi_hip.h - C-ABI of the MI355X-native MPPI rollout backend (libmppi_hip.so).
 *
 * The reference (tud-airlab/mppi-isaac) has no FFI: its hot path sits behind Python
 * callables bound to two absent native engines (Isaac Gym / PhysX and mppi_torch).
 * Each entry point below names the reference interface it replaces (paths relative
 * to the reference tree).  Host code (Python, mppi-isaac_amd/mppiisaac/backend/capi.py)
 * binds these with ctypes; no torch types cross this boundary - only plain pointers,
 * sizes and an opaque hipStream_t.
 *
 * Conventions
 *  - All model/config parameters are double; device arithmetic is fp32.
 *  - Device buffers are sample-minor ("SoA"): element (row r, sample k) of a K-wide
 *    buffer lives at r*K + k, so a wavefront reads 64 consecutive floats.
 *  - Reference-layout ("AoS", env-major) tensors exist only at the materialise
 *    boundary (mppi_sim_materialise), as the reference's gym state tensors do
 *    (mppiisaac/planner/isaacgym_wrapper.py:186-199).
 *  - Every function returns 0 on success, a negative MPPI_E* code on failure;
 *    mppi_last_error() returns the message of the last failure in this thread.
 *  - One context per (process, device); contexts are not thread-safe.
 */
#ifndef MPPI_HIP_H
#define MPPI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPPI_ABI_VERSION 9

#define MPPI_MAX_BODIES 12   /* moving bodies (= DOF) of the articulated robot        */
#define MPPI_MAX_LINKS 24    /* reported rigid bodies of the robot (URDF links)      */
#define MPPI_MAX_ACTORS 12   /* actors per env (robot + boxes/spheres; reference IsaacGymConfig.num_obstacles = 10, isaacgym_wrapper.py:16) */
#define MPPI_MAX_NU 12       /* control dimension                                    */
#define MPPI_MAX_H 64        /* horizon                                              */
#define MPPI_MAX_KNOTS 16    /* spline knots of the halton-spline sampler            */
#define MPPI_MAX_COST_W 16
#define MPPI_MAX_SHAPES 64   /* collision primitives per env (anymal: 37; an arm + ten obstacle spheres)  */
#define MPPI_MAX_PAIRS 128   /* candidate contact pairs per env (one verdict bit per pair in four mask words; the ten-link arm
                              * among the reference's ten obstacle spheres - IsaacGymConfig.num_obstacles - has 100)        */
#define MPPI_CONTACT_POINT_NORMALS 1
#define MPPI_CONTACT_EXPLICIT_LIGHT 2   /* contact_flags bit 1, see mppi_model_t */
/* ABI 9: a free actor of at most MPPI_LIGHT_BODY_MASS kg that is at least MPPI_LIGHT_BODY_RATIO times lighter than the robot whose link it
 * touches is held IMPLICITLY (mppi_model_t.contact_flags); the ramp depth of such a pair is contact_ramp_depth / MPPI_LIGHT_RAMP_DIV */
#define MPPI_LIGHT_BODY_MASS 0.25
#define MPPI_LIGHT_BODY_RATIO 100
#define MPPI_LIGHT_RAMP_DIV 16
/* free actors: the angular velocity of a free rigid body is limited to this [rad/s] at the end of every substep - Isaac Gym's
 * AssetOptions.max_angular_velocity, whose default the reference keeps (isaacgym_utils.py:15: gymapi.AssetOptions() as it comes;
 * PhysX clamps every rigid body of the asset).  A one-gram block pinched at an angle between two gripper pads leaves like a squeezed
 * seed, at metres per second and hundreds of rad/s; at h = 25 ms that is more than a radian per substep, and every contact it meets
 * afterwards is evaluated at a pose that is gone by the end of the substep */
#define MPPI_MAX_ANGULAR_VELOCITY 64.0
#define MPPI_MAX_FREE 4      /* free (non-fixed) box/sphere actors per env (the shipped kernels carry 2 slots; scenes with
                              * 3-4 free actors get their kernels built on demand, see mppi_create)          */
#define MPPI_MAX_EXTRA_BASES 3 /* moving-base robots per env beyond the first (ABI 7)         */

enum { MPPI_OK = 0, MPPI_EINVAL = -1, MPPI_EHIP = -2, MPPI_EUNSUPPORTED = -3, MPPI_ESTATE = -4 };
enum { MPPI_JOINT_REVOLUTE = 0, MPPI_JOINT_PRISMATIC = 1 };
/* dof_mode of the reference's ActorWrapper (isaacgym_wrapper.py:52, drive gains :491-507; position: see mppi_model_t.drive_kp) */
enum { MPPI_DRIVE_VELOCITY = 0, MPPI_DRIVE_EFFORT = 1, MPPI_DRIVE_POSITION = 2 };
/* ActorWrapper.type (isaacgym_wrapper.py:42-46, isaacgym_utils.py:19-54) */
enum { MPPI_ACTOR_ROBOT = 0, MPPI_ACTOR_BOX = 1, MPPI_ACTOR_SPHERE = 2 };
/* fused stage costs: restatements of the reference's example Objectives */
enum {
    MPPI_COST_NONE = 0,        /* generic mode: cost comes from the host callback          */
    MPPI_COST_POINT_REACH = 1, /* benchmarks/point_robot/mppi_planner/mppi_planner_wrapper.py:17-35 (nav term) */
    MPPI_COST_PANDA_REACH = 2, /* examples/panda/planner.py:22-40                           */
    MPPI_COST_BOXER_PUSH = 3,  /* examples/boxer_push/planner.py:26-67                      */
    MPPI_COST_PANDA_PICK = 4,  /* examples/panda_pick/planner.py:24-53                      */
    MPPI_COST_PROGRAM = 5      /* a list of mppi_term_t: every Objective of examples/<x>/planner.py is a weighted sum of the
                                * measurements below; evaluated inside the rollout kernels (contact scenes: the octet kernel;
                                * contact-free scenes: the one-lane kernel)                                             */
};
/* measurements of a cost program (one value per env; mppiisaac/objectives.py holds the same algebra for generic mode) */
enum {
    MPPI_OP_DIST = 1,     /* || a[:n] - b[:n] ||                                                              */
    MPPI_OP_TILT = 2,     /* size of the first two "ZYX" Euler angles of body a's quaternion read xyzw-as-(r,i,j,k), as the
                           * reference's arm objectives do (examples/panda/planner.py:30-32)                    */
    MPPI_OP_YAW_ABS = 3,  /* | yaw(quaternion of actor a) - p[3] |   (mppiisaac/utils/conversions.py:4-11)      */
    MPPI_OP_ALIGN = 4,    /* 1 + cos of the planar angle at b between the rays to a and to c                   */
    MPPI_OP_FORCE_L1 = 5, /* sum_j<n |net contact force of rigid body idx[0]|_j                                */
    MPPI_OP_SPEED = 6,    /* || linear velocity of actor a [:n] ||                                             */
    MPPI_OP_DOF_SQ = 7,   /* sum_{i = idx[0]}^{idx[1]-1} (x_i - p[i - idx[0]])^2, x = DOF positions (n = 0) / velocities (n = 1);
                           * idx[2] = number of reference values given in p (0: none)                           */
    MPPI_OP_ABS_DZ = 8,   /* | a.z - b.z |                                                                     */
    MPPI_OP_BELOW = 9     /* max(p[3] - a.z, 0)                                                                */
};
/* where an operand's 3-vector comes from */
enum {
    MPPI_SRC_NONE = 0,
    MPPI_SRC_RB = 1,      /* position (TILT: orientation) of rigid body idx - a robot link or a box / sphere body */
    MPPI_SRC_ACTOR = 2,   /* root row of actor idx (position; SPEED: linear velocity; YAW_ABS: quaternion)      */
    MPPI_SRC_DOF_XY = 3,  /* (q_0, q_1, 0): the planar base of the point robot                                 */
    MPPI_SRC_CONST = 4    /* the constant (p[0], p[1], p[2])                                                   */
};
#define MPPI_MAX_TERMS 16
typedef struct mppi_term {
    int32_t op;        /* MPPI_OP_*                                                                     */
    int32_t n;         /* number of components (DIST, FORCE_L1, SPEED); DOF_SQ: 0 positions, 1 velocities */
    int32_t src[3];    /* MPPI_SRC_* of the operands a, b, c                                            */
    int32_t idx[3];    /* their rigid-body / actor indices (DOF_SQ, FORCE_L1: see the op)               */
    double w;          /* weight of the term                                                            */
    double p[8];       /* constants (see the ops)                                                       */
} mppi_term_t;
/* noise sources of the sampler (mppi_mode / sampling_method of the conf/mppi files):
 *   HALTON_SPLINE  fixed low-discrepancy set: scrambled-Halton knots -> Phi^-1 -> B-spline (mppi_sample, once)
 *   EXTERNAL       the caller owns a device buffer eps [H][nu][K] (mppi_set_noise_dev)
 *   NORMAL         counter-based Gaussian draws (Philox4x32-10 keyed by seed, counter = global sample id, control
 *                  dimension, knot block, iteration; Box-Muller in fp64), redrawn by mppi_sample_normal every
 *                  control iteration: knots through spline_basis ("halton-spline" mode with sampling_method
 *                  "random"), or one draw per horizon step when n_knots == horizon (mppi_mode "simple")       */
enum { MPPI_SAMPLE_HALTON_SPLINE = 0, MPPI_SAMPLE_EXTERNAL = 1, MPPI_SAMPLE_NORMAL = 2 };
/* collision primitives: URDF <collision> boxes / meshes (as their AABB) / spheres / thin cylinders
 * (wheels, casters: "disc", axis = local z of the shape frame); box and sphere actors
 * (isaacgym_utils.py:26-52).  Contact model: DESIGN.md section 3 (build-normative). */
enum { MPPI_SHAPE_BOX = 0, MPPI_SHAPE_SPHERE = 1, MPPI_SHAPE_DISC = 2 };

typedef struct mppi_shape {
    int32_t actor;     /* owning actor                                                   */
    int32_t body;      /* robot shapes: moving body it is welded to, -1 = robot base; others: -1 */
    int32_t type;      /* MPPI_SHAPE_*                                                   */
    int32_t rb;        /* rigid-body row that receives its net contact force             */
    double size[3];    /* box: half extents; sphere/disc: radius in size[0]              */
    double R[9];       /* shape frame in the body (or actor) frame                       */
    double p[3];
    double friction;   /* per-shape friction (isaacgym_wrapper.py:467-480; casters 0)    */
} mppi_shape_t;

typedef struct mppi_pair {
    int32_t a, b;      /* shape indices; b = -1: ground plane z = 0                      */
} mppi_pair_t;

/* One moving body = one 1-DOF joint + the links welded to its child link.
 * Transform convention: x_parent = R * x_child + p (R row-major 3x3). */
typedef struct mppi_body {
    int32_t parent;      /* moving-body index, -1 = robot base (-1 - r: base r of a forest, ABI 7) */
    int32_t jtype;       /* MPPI_JOINT_*                                                 */
    double axis[3];      /* unit joint axis in the body (child) frame                    */
    double R_tree[9];    /* joint frame in the parent body frame at q = 0                */
    double p_tree[3];
    double mass;         /* composite rigid inertia about the body-frame origin:        */
    double h[3];         /*   first moment m*c                                           */
    double Io[6];        /*   xx xy xz yy yz zz                                          */
    int32_t limited;     /* joint position limits active (URDF revolute/prismatic)       */
    int32_t pad_;
    double lower, upper; /* URDF <limit lower upper>                                     */
    double effort;       /* URDF <limit effort>: drive force clamp, 0 = none             */
    double velocity;     /* URDF <limit velocity>: |qdot| clamp, 0 = none                */
} mppi_body_t;

/* A reported rigid body (URDF link): welded to `body` (-1 = base) at (R,p). */
typedef struct mppi_link {
    int32_t body;
    int32_t pad_;
    double R[9];
    double p[3];
} mppi_link_t;

/* conf/actors/<name>.yaml -> ActorWrapper (isaacgym_wrapper.py:49-77) */
typedef struct mppi_actor {
    int32_t type;      /* MPPI_ACTOR_*        */
    int32_t fixed;     /* fix_base_link       */
    int32_t collision; /* collision group on  */
    int32_t gravity;   /* !disable_gravity    */
    double size[3];
    double mass;
    double friction;
    int32_t first_rb;  /* first rigid-body row of this actor in rigid_body_state         */
    int32_t n_rb;      /* robot: n_links, box/sphere: 1                                  */
    /* per-env randomisation of box/sphere actors (isaacgym_wrapper.py:430-475, isaacgym_utils.py:30-52) */
    double noise_sigma_size[3];    /* size += N(0, sigma)                                */
    double noise_percentage_mass;  /* mass += U(-p, p) * mass                            */
    double noise_percentage_friction; /* friction += U(-p, p) * friction                 */
} mppi_actor_t;

/* Scene of one env: one articulated robot - or a forest of robots, fixed-base (merged into one tree set hanging off the world)
 * or moving-base (one floating base per tree, ABI 7 fields at the end) - + simple actors, in env_cfg (= root_state) order. */
typedef struct mppi_model {
    int32_t abi_version;
    int32_t n_actors;
    mppi_actor_t actors[MPPI_MAX_ACTORS];
    int32_t robot_actor; /* index of the robot in the actor list                          */
    int32_t n_bodies;    /* = number of DOF                                               */
    mppi_body_t bodies[MPPI_MAX_BODIES];
    int32_t n_links;
    int32_t n_rb;        /* total rigid bodies per env (all actors)                       */
    mppi_link_t links[MPPI_MAX_LINKS];
    double base_mass;    /* composite inertia of the root-link cluster (floating base)    */
    double base_h[3];
    double base_Io[6];
    int32_t drive_mode;  /* MPPI_DRIVE_*                                                  */
    int32_t substeps;    /* conf/isaacgym/<x>.yaml                                        */
    double drive_kd;     /* 600 velocity / 10 effort / 0 position (isaacgym_wrapper.py:491-504)  */
    /* MPPI_DRIVE_POSITION (isaacgym_wrapper.py:501-504,571-572): apply_robot_cmd OVERWRITES the DOF state with the command -
     * q <- target, qdot <- 0 at the start of every simulator step - and the step then runs with the position drive
     * tau = kp (target - q) - kd qdot, kp = drive_kp = 80 (treated implicitly like the other drives).  The reference never sets a
     * position target (it would stay 0: a spring to zero behind a teleport to u); the drive here holds the commanded pose. */
    double drive_kp;
    double dt;
    double gravity[3];   /* (0,0,-9.8) isaacgym_wrapper.py:29                             */
    /* command scatter of apply_robot_cmd (isaacgym_wrapper.py:524-572), folded into at
     * most two (column, coefficient) terms per DOF: target_i = c0*u[col0] + c1*u[col1].
     * Plain DOFs use (col,1),(0,0); diff-drive wheels use (0,1/r),(1,-+L/2r) (_ik :510-522). */
    int32_t nu;
    int32_t cmd_col[MPPI_MAX_BODIES][2];
    double cmd_coef[MPPI_MAX_BODIES][2];
    /* contact scene (all zero for contact-free scenes) */
    int32_t n_shapes;
    int32_t n_pairs;
    mppi_shape_t shapes[MPPI_MAX_SHAPES];
    mppi_pair_t pairs[MPPI_MAX_PAIRS];
    double ground_friction;   /* add_ground_plane: 1.0 (isaacgym_utils.py:61-68)              */
    double contact_alpha;     /* penalty stiffness  k = alpha * m_eff / h^2 per contact patch */
    double contact_beta;      /* normal damping     c = beta  * m_eff / h                     */
    double friction_beta;     /* stick damping      c_t = friction_beta * m_eff / h           */
    /* depth over which the velocity-proportional part of the normal force (damper + implicit end-of-step spring term)
     * ramps in linearly, Hunt-Crossley style: the contact force is continuous at touch-down instead of jumping by
     * (c + k h) v_n.  0 = full strength from the first touch (the law of ABI 1).  Host default: the static sag |g| h^2 / alpha */
    double contact_ramp_depth;
    /* >= 0: every sample (global id g) draws its own box sizes / masses / frictions from a counter-based
     * hash of (seed, g, actor) - the seeded counterpart of the reference's unseeded np.random draws per env;
     * < 0: nominal values in every sample */
    int32_t randomize_seed;
    /* bit 0 (MPPI_CONTACT_POINT_NORMALS): two DYNAMIC boxes keep the law of ABI <= 7 - every feature point pushed towards the nearest
     * face of the other box.  Default (0): ONE normal per pair from the 15-axis separating-axis test, every point's depth measured
     * along it, and a patch its points under-sample (a lone corner, crossing edges) filled up to half the nominal stiffness by one
     * contact of the separating-axis depth
     * bit 1 (MPPI_CONTACT_EXPLICIT_LIGHT), ABI 9: a LIGHT free actor (MPPI_LIGHT_BODY_MASS, MPPI_LIGHT_BODY_RATIO) against a robot link keeps the
     * explicit law of two dynamic bodies, whose stiffness alpha m / h^2 is what the lighter body can carry in an explicit step - 1.3 N/m
     * for the 1-gram block of the reference's conf/actors/panda_pick_block.yaml: a gripper closes through it.  Default (0): such a pair
     * is implicit on BOTH bodies with the robot's gains - the link sees the light body as a wall moving with its start-of-substep
     * velocity, the light body is solved after the robot against the link's end-of-substep velocity (DESIGN.md 3 "light bodies") -
     * what PhysX's implicit solver does for examples/panda_pick and examples/omni_panda_pick (isaacgym_wrapper.py:29-36) */
    int32_t contact_flags;
    /* ABI 7 - several MOVING-base robots in one env (reference isaacgym_wrapper.py:101-106,534-559, conf/mppi/multi-jackal.yaml).
     * The robots form one articulated forest (bodies, links and DOFs follow one another in env order, like the fixed-base
     * forests); base 0 is `robot_actor` with base_mass / base_h / base_Io above, base r > 0 is extra_base_*[r - 1].  A body whose
     * `parent`, a link or a shape whose `body` is -1 - r hangs off base r (r = 0: the -1 of every single-robot model).  All
     * robots of an env are either fixed or moving.  Every base has its own root row (root_state[actor]) and its own 6x6 base
     * system in the articulated-body solve; pairs between shapes of DIFFERENT trees are allowed (the host lists chassis against
     * chassis: round 5), pairs within one tree are not. */
    int32_t n_extra_bases;
    int32_t extra_base_actor[MPPI_MAX_EXTRA_BASES];
    double extra_base_mass[MPPI_MAX_EXTRA_BASES];
    double extra_base_h[MPPI_MAX_EXTRA_BASES][3];
    double extra_base_Io[MPPI_MAX_EXTRA_BASES][6];
} mppi_model_t;

/* mppi_torch.MPPIConfig fields (reference conf/mppi/ + benchmarks/point_robot/setup/mppi.yaml:5-37) */
typedef struct mppi_config {
    int32_t abi_version;
    int32_t num_samples;   /* K: samples owned by THIS context (one shard)               */
    int32_t horizon;       /* H                                                          */
    int32_t nu;
    int32_t k_offset;      /* global id of local sample 0 (sharding)          */
    int32_t k_total;       /* global number of samples over all shards                   */
    int32_t sample_null_action; /* global sample k_total-1 uses u = 0                    */
    int32_t use_priors;    /* global sample k_total-2 uses the prior sequence            */
    int32_t sampling;      /* MPPI_SAMPLE_*                                              */
    int32_t n_knots;       /* knots per control dim (H/4; = H if < 3); NORMAL: = H -> no spline */
    int32_t noise_abs_cost;
    int32_t want_rollouts; /* record visualize_link positions [H][K][3] (get_rollouts)   */
    int32_t viz_link;      /* link index recorded when want_rollouts                     */
    int32_t seed;
    double lambda_;
    double rollout_var_discount; /* gamma                                                */
    double u_init;
    double u_min[MPPI_MAX_NU], u_max[MPPI_MAX_NU];
    double noise_sigma_diag[MPPI_MAX_NU]; /* diagonal of noise_sigma (variances)        */
    double noise_mu[MPPI_MAX_NU];         /* mean of the sampled noise (MPPIConfig.noise_mu; NORMAL sampling) */
    double spline_basis[MPPI_MAX_H * MPPI_MAX_KNOTS]; /* [H][n_knots] row-major          */
} mppi_config_t;

typedef struct mppi_cost {
    int32_t kind;          /* MPPI_COST_*                                                */
    int32_t link[4];       /* rigid-body / link indices used by the cost                 */
    int32_t actor[6];      /* actor indices used by the cost                             */
    double w[MPPI_MAX_COST_W]; /* weights / scalar parameters            */
    int32_t n_terms;       /* MPPI_COST_PROGRAM: cost = sum_i terms[i].w * measurement_i        */
    int32_t pad_;
    mppi_term_t terms[MPPI_MAX_TERMS];
} mppi_cost_t;

typedef struct mppi_ctx mppi_ctx_t;

const char *mppi_last_error(void);
int mppi_abi_version(void);
int mppi_device_count(int *count);

/* ---- lifetime: replaces IsaacGymWrapper.__init__/start_sim (isaacgym_wrapper.py:84-236)
 *      + MPPIPlanner(cfg.mppi, ...) construction (mppi_isaac.py:43-49).  Allocates all
 *      device buffers on `device`.  `stream` is a hipStream_t (NULL = default stream).   */
/*      Kinematic trees: the rollout kernels are templates over the tree.  The library ships the instantiations of the robots under
 *      assets/compiled/; ANY OTHER tree (what gym.load_asset accepts: whatever URDF the actor YAML names, isaacgym_utils.py:14-29)
 *      is built on demand at mppi_create - hipcc compiles the two generated units of that tree into a plugin library that is cached
 *      on disk (MPPI_JIT_CACHE, default ~/.cache/mppi_hip; keyed by tree, flags and the kernel sources), loaded and appended to the
 *      launch table; so are the contact-scene kernels with four free-actor slots for envs of 3-4 free actors.  MPPI_JIT=0 switches
 *      the builds off (unknown trees are refused with MPPI_EUNSUPPORTED), HIPCC names the compiler.  mppi_jit_info: what the last
 *      on-demand build did ("built <plugin> in 34.1 s" / "cached <plugin> in 0.0 s"; empty string: none so far).            */
int mppi_create(const mppi_model_t *model, const mppi_config_t *cfg, int device, mppi_ctx_t **out);
int mppi_jit_info(char *buf, int buflen);
int mppi_destroy(mppi_ctx_t *ctx);
int mppi_set_stream(mppi_ctx_t *ctx, void *hip_stream);
int mppi_synchronize(mppi_ctx_t *ctx);

/* ---- state in: replaces reset_rollout_sim (mppi_isaac.py:87-105): ONE env state
 *      (dof_state [2n] interleaved q,qdot; root_state [A][13] pos,quat xyzw,linvel,angvel)
 *      is broadcast to all K samples inside the kernel (no [K,..] copy is written). */
/*      (ABI 8: the host state rides as kernel arguments of a one-wavefront launch - stream-ordered, no copy operation, no
 *      synchronise; the host buffers are free again when the call returns.) */
int mppi_set_state(mppi_ctx_t *ctx, const float *dof_state_host, const float *root_state_host);
int mppi_set_state_dev(mppi_ctx_t *ctx, const float *dof_state_dev, const float *root_state_dev);
int mppi_get_state(mppi_ctx_t *ctx, float *dof_state_host, float *root_state_host);

/* ---- MPPI core: replaces mppi_torch.MPPIPlanner.command (call sites mppi_isaac.py:84,113) */
int mppi_set_cost(mppi_ctx_t *ctx, const mppi_cost_t *cost);
int mppi_sample(mppi_ctx_t *ctx, uint32_t index_base);        /* halton-spline -> eps [H][nu][K] */
/* NORMAL sampling: eps = noise_mu + sqrt(noise_sigma) * (basis . z), z ~ N(0,1) from Philox4x32-10 with key
 * (config.seed, 'MPPI') and counter (global sample id, control dim, knot block, iteration): the draw depends on the
 * GLOBAL sample id only, so any sharding sees the same noise.  Call once per control iteration (what mppi_torch's
 * simple mode does with torch's generator inside MPPIPlanner.command, call sites mppi_isaac.py:84,113). */
int mppi_sample_normal(mppi_ctx_t *ctx, uint32_t iteration);
int mppi_set_noise_dev(mppi_ctx_t *ctx, const float *eps_dev); /* external eps [H][nu][K]        */
int mppi_set_prior(mppi_ctx_t *ctx, const float *prior_host);  /* [H][nu] for sample k_total-2   */
int mppi_set_prior_row(mppi_ctx_t *ctx, int t, const float *row_host); /* [nu]: prior(state, t) evaluated AT rollout step t (generic mode) */
int mppi_set_nominal(mppi_ctx_t *ctx, const float *U_host);    /* [H][nu]                        */
int mppi_get_nominal(mppi_ctx_t *ctx, float *U_host);
/* filter_u: linear smoothing operator F [H][H] applied to the updated nominal (U <- F U) before the action is taken;
 * NULL switches it off.  The host builds F as a Savitzky-Golay filter (mppiisaac/planner/mppi.py:savgol_matrix). */
int mppi_set_filter(mppi_ctx_t *ctx, const float *F_host);
int mppi_rollout(mppi_ctx_t *ctx);   /* persistent kernel: K samples x H steps, fused cost -> S[K], du */
int mppi_reduce(mppi_ctx_t *ctx, float *record_out_dev); /* shard record (beta, eta, N[H*nu]); NULL = internal buffer */
int mppi_record_floats(const mppi_ctx_t *ctx);               /* 2 + H*nu                             */
/* The fused quad rollout kernels fold their per-wavefront records per XCD group in their own tail: after mppi_rollout the
 * shard is described by mppi_shard_record_count() records of mppi_record_floats() floats (8; 1 for grids that are no
 * multiple of 16 wavefronts; 0 = this context does not fold - lane kernels, generic mode - use mppi_reduce).
 * mppi_set_record_out points the fold at a caller buffer [count][2+H*nu] (NULL = internal), e.g. this rank's rows of the
 * tensor that is all-gathered in place: the sharded iteration is then rollout -> all-gather -> mppi_update(all records). */
int mppi_shard_record_count(const mppi_ctx_t *ctx);
int mppi_set_record_out(mppi_ctx_t *ctx, float *records_dev);
int mppi_record_dev(mppi_ctx_t *ctx, float **record_dev);    /* device pointer of this shard's record */
/* Direct exchange of the shard records between the GPUs of one node, owned by the library ("each rank stores its
 * record into a mailbox on every peer, then flag / poll" - the 1-hop all-gather over the fully connected xGMI mesh; the
 * reference has no counterpart, isaacgym_wrapper.py:126 is single-GPU).  The alternative to all-gathering the records with
 * RCCL: no collective library, no host thread per iteration, two small kernels that live in a captured graph.
 *   mppi_mailbox_create   after mppi_set_cost: this rank's inbox (n_ranks flags + 2 x n_ranks slots of
 *                         max(1, mppi_shard_record_count) records), in fine-grained device memory where available
 *   mppi_mailbox_ptr / mppi_mailbox_ipc_handle   what peers need: the device pointer (same process) or a 64-byte hipIpc handle
 *   mppi_mailbox_set_peer / mppi_mailbox_open    connect rank `peer_rank`'s inbox by pointer / by IPC handle (own rank: no-op)
 *   mppi_exchange         after mppi_rollout, on the context's stream: publish own records to every inbox, wait (bounded: ~2 s)
 *                         for every rank's records of this iteration, copy them to the fixed buffer of mppi_mailbox_gathered
 *   mppi_mailbox_gathered [n_ranks * count][2+H*nu] device buffer (fixed address) and its record count, for mppi_update /
 *                         mppi_update_step_world
 *   mppi_exchange_publish / mppi_exchange_wait   the two halves of mppi_exchange.  A caller that drives several ranks of ONE
 *                         device from one thread (tests) enqueues every rank's publish before any rank's wait: a waiting
 *                         kernel occupies its hardware queue, and streams of one process may share hardware queues
 *   mppi_exchange_status  1 if a wait timed out (a peer never published) SINCE THE LAST CALL of this function (read-and-clear,
 *                         ABI 8: a transient stall is reported once, for the iterations it affected).  The records of a rank
 *                         that was late are NEUTRAL in the gathered buffer of that iteration (eta = 0: the update runs on the
 *                         ranks that did publish, never on stale or half-written records); callers poll the status and warn
 *   mppi_mailbox_info     whether the inbox is fine-grained device memory (required for peers of another process / GPU:
 *                         mppi_mailbox_ipc_handle refuses a coarse-grained inbox), records per rank, number of ranks
 * A shard whose rollout folded no records (generic Objective mode, ragged grids) publishes ONE reduced record and neutral padding
 * whatever the mailbox was sized for; mppi_mailbox_create is failure-atomic (a failed call leaves no mailbox behind). */
int mppi_mailbox_create(mppi_ctx_t *ctx, int rank, int n_ranks);
int mppi_mailbox_info(mppi_ctx_t *ctx, int *fine_grained, int *records_per_rank, int *n_ranks);
int mppi_mailbox_ptr(mppi_ctx_t *ctx, void **inbox_dev, size_t *bytes);
int mppi_mailbox_ipc_handle(mppi_ctx_t *ctx, void *handle64);
int mppi_mailbox_set_peer(mppi_ctx_t *ctx, int peer_rank, void *inbox_dev);
int mppi_mailbox_open(mppi_ctx_t *ctx, int peer_rank, const void *handle64);
int mppi_mailbox_gathered(mppi_ctx_t *ctx, float **records_dev, int *n_records);
int mppi_exchange(mppi_ctx_t *ctx);
int mppi_exchange_publish(mppi_ctx_t *ctx);
int mppi_exchange_wait(mppi_ctx_t *ctx);
int mppi_exchange_status(mppi_ctx_t *ctx, int *timed_out);
/* mppi_exchange followed by mppi_update_step_world(planner, gathered records, world) - as ONE launch where that is possible
 * (fixed-base contact-free scenes whose rollout leaves per-wavefront records, one shard record per rank in the mailbox: the
 * combine + world kernel starts with the reduction to the shard record, the publish and the bounded wait), so that the sharded
 * control iteration has the two launches of the unsharded one; every other context takes the two calls in turn. */
int mppi_exchange_update_step_world(mppi_ctx_t *planner, mppi_ctx_t *world);
/* combine n shard records (device, [n][2+H*nu]; NULL = own record), update U, emit action, shift */
int mppi_update(mppi_ctx_t *ctx, const float *records_dev, int n_records);
int mppi_get_action(mppi_ctx_t *ctx, float *action_host);     /* [nu]; synchronises the stream        */
/* The action of the LAST update as soon as the update kernel has published it (the kernel mirrors the action and a
 * sequence number into mapped pinned host memory; this call polls the number): returns before the stream is idle,
 * e.g. while the fused K=1 world step of mppi_update_step_world still runs.  Work enqueued afterwards stays ordered
 * behind it on the stream.  Same role as the `.cpu()` of the action in mppi_isaac.py:84. */
int mppi_wait_action(mppi_ctx_t *ctx, float *action_host);
int mppi_action_dev(mppi_ctx_t *ctx, float **action_dev);
/* n updates (mppi_update / mppi_update_step_world) ran on the device WITHOUT a call of this API - replays of a HIP graph
 * the launches were captured into (n = +1 per replay) - or were recorded by this API but never ran - the launches made
 * while the stream was being captured (n = -1 each): keeps the sequence number mppi_wait_action waits for in step */
int mppi_note_graph_update(mppi_ctx_t *ctx, int n);
int mppi_command(mppi_ctx_t *ctx, float *action_host);        /* rollout+reduce+update+get_action     */
int mppi_get_costs(mppi_ctx_t *ctx, float *S_host);           /* [K] total trajectory costs           */
int mppi_get_weights_stats(mppi_ctx_t *ctx, float *beta_eta_host); /* [2]                             */
/* mppi_torch's `update_lambda` (MPPIConfig.update_lambda / eta_u_bound / eta_l_bound; False in every shipped conf): the caller
 * reads eta with mppi_get_weights_stats after an update and sets the temperature of the NEXT iteration (stream-ordered; takes
 * effect for the next rollout's control cost and the next update's weights).  lambda > 0. */
int mppi_set_lambda(mppi_ctx_t *ctx, double lambda);
int mppi_get_rollouts(mppi_ctx_t *ctx, float *viz_host);      /* [H][K][3], get_rollouts mppi_isaac.py:118-124 */
int mppi_get_perturbations(mppi_ctx_t *ctx, float *du_host);  /* [H][nu][K] effective perturbations   */
int mppi_get_noise(mppi_ctx_t *ctx, float *eps_host);         /* [H][nu][K]                           */

/* ---- parity / debug: the context's MPPI_COST_PROGRAM evaluated on the device by the interpreter the rollout kernels run, on
 *      CALLER-GIVEN simulator answers of n envs (host arrays in the reference layouts: dof [n][2*n_dof], root [n][A][13],
 *      rb [n][B][13], cf [n][B][3]) instead of the kernel's own kinematics -> cost [n].  What Objective.compute_cost(sim) is to
 *      the gym getters (reference examples/<x>/planner.py compute_cost over isaacgym_wrapper.py:238-330): the golden Objective
 *      fixtures are pushed through the HIP cost path with it (tests/test_gpu_parity.py). */
int mppi_eval_cost(mppi_ctx_t *ctx, int n, const float *dof_host, const float *root_host, const float *rb_host, const float *cf_host, float *cost_host);

/* ---- batched simulator (generic Objective mode and the K=1 "world"):
 *      replaces IsaacGymWrapper.apply_robot_cmd + step (isaacgym_wrapper.py:524-572,639-655)
 *      and the four gym state tensors (:186-199). */
int mppi_sim_reset(mppi_ctx_t *ctx);                       /* all K envs <- current x0              */
int mppi_sim_step(mppi_ctx_t *ctx, const float *u_dev, int u_is_shared); /* u [K][nu] (AoS) or [nu] */
int mppi_sim_step_horizon(mppi_ctx_t *ctx, int t);         /* u = clamp(U[t]+eps[t]) per sample     */
/* (ABI 8) the host side of the reference's world loop (examples/<x>/world.py:35-44: torch_to_bytes(sim._dof_state),
 * torch_to_bytes(sim._root_state) -> planner -> sim.apply_robot_cmd(action); sim.step()) without copy operations:
 *   mppi_sim_step_host   one command [nu] for every env by HOST pointer (written into a ring slot of the context's mapped host
 *                        block; the step kernel reads it through the mapped pointer)
 *   mppi_sim_materialise_mirror   K = 1: mppi_sim_materialise whose kernel also writes env 0's dof [2n] / root [A][13] rows into
 *                        the mapped host block and publishes a sequence number behind them
 *   mppi_mirror_wait     polls that number and copies the mirrored state out: no device-to-host copy, no stream synchronise */
int mppi_sim_step_host(mppi_ctx_t *ctx, const float *u_host);
int mppi_sim_materialise_mirror(mppi_ctx_t *ctx, float *dof_dev, float *root_dev, float *rb_dev, float *cf_dev);
int mppi_mirror_wait(mppi_ctx_t *ctx, float *dof_host, float *root_host);
/* reference-layout tensors: dof [K][2n], root [K][A][13], rb [K][B][13], cf [K][B][3]; NULL = skip */
int mppi_sim_materialise(mppi_ctx_t *ctx, float *dof_dev, float *root_dev, float *rb_dev, float *cf_dev);
int mppi_sim_accumulate_cost(mppi_ctx_t *ctx, int t, const float *cost_dev); /* S += gamma^t c      */
int mppi_sim_finish(mppi_ctx_t *ctx);                      /* S += control cost                     */
/* The same horizon in TWO launches for host-side costs that are functions of the state tensors (mppi_isaac.py:57-69 evaluates
 * running_cost once per horizon step; the dynamics never depend on it): the fused rollout kernel with MPPI_COST_NONE and
 * every step's env state kept (S = control cost, du as in mppi_rollout), then the reference-layout tensors of all H*K
 * env-steps - dof [H*K][2n], root [H*K][A][13], rb [H*K][B][13], cf [H*K][B][3], row t*K + k = env k after step t; NULL =
 * skip.  Costs evaluated on those rows are added with mppi_sim_accumulate_cost(ctx, 0, sum_t gamma^t c_t).
 * MPPI_EUNSUPPORTED for contexts that run the one-lane kernels or a contact scene with fewer than 8 samples. */
int mppi_rollout_trajectory(mppi_ctx_t *ctx);
int mppi_materialise_trajectory(mppi_ctx_t *ctx, float *dof_dev, float *root_dev, float *rb_dev, float *cf_dev);
/* (ABI 8) one rigid body of all H*K env-steps as dense rows [H*K][13]: what `sim.get_actor_link_by_name(actor, link)` of an
 * Objective reads (reference isaacgym_wrapper.py:302-308) without the whole [H*K][B][13] tensor.  MPPI_EUNSUPPORTED for contact
 * scenes and for rigid bodies that are no robot link: callers take the full tensor. */
int mppi_materialise_trajectory_link(mppi_ctx_t *ctx, int rb_index, float *out_dev);
/* (ABI 8) ... and the other end of that horizon in ONE launch: cost_dev [H][K] = the stage costs of all env-steps as the host-side
 * Objective returned them (row t*K + k); S_k += sum_t gamma^t c_t[k] + control cost, then the per-wavefront records (what
 * mppi_sim_accumulate_cost, mppi_sim_finish and mppi_reduce do in three); record_out_dev as in mppi_reduce.  MPPI_EUNSUPPORTED
 * for contexts of the one-lane kernels. */
int mppi_reduce_horizon_costs(mppi_ctx_t *ctx, const float *cost_dev, float *record_out_dev);
/* device-resident closed loop: step a K=1 world with the planner's action, feed its state back */
int mppi_world_step_from(mppi_ctx_t *world, mppi_ctx_t *planner);
int mppi_set_state_from_world(mppi_ctx_t *planner, mppi_ctx_t *world);
/* mppi_update + mppi_world_step_from + mppi_set_state_from_world; one kernel for fixed-base contact-free scenes */
int mppi_update_step_world(mppi_ctx_t *planner, const float *records_dev, int n_records, mppi_ctx_t *world);

/* ---- instrumentation (reference has only print(FPS), examples/panda/world.py:53-59) */
int mppi_set_profiling(mppi_ctx_t *ctx, int on);            /* hipEvent brackets on the context's stream: 0 off, n >= 1 every n-th launch */
int mppi_kernel_ms(mppi_ctx_t *ctx, int which, float *ms); /* mean launch duration since profiling was enabled: 0 rollout 1 reduce 2 update */
int mppi_kernel_info(mppi_ctx_t *ctx, char *buf, int buflen);
/* per-wavefront residency of the quad rollout kernels: start / end of every wavefront of the LAST rollout in ticks of the
 * 100 MHz constant clock (s_memrealtime), [n_wavefronts][2] - load-balance evidence (tools/exp/wave_balance.py) */
int mppi_set_wave_clock(mppi_ctx_t *ctx, int on);
int mppi_get_wave_clock(mppi_ctx_t *ctx, uint64_t *start_end_host, int n_wavefronts);

#ifdef __cplusplus
}
#endif
#endif /* MPPI_HIP_H */
