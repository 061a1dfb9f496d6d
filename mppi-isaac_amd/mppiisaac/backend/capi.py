"""ctypes binding of include/mppi_hip.h (libmppi_hip.so).

The structures mirror the header field by field; `check_abi()` compares sizeof() on
both sides when the library is loaded.  There is NO CPU fallback here: if the HIP
library is missing or does not load, importing the product path fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

MAX_BODIES, MAX_LINKS, MAX_ACTORS, MAX_NU, MAX_H, MAX_KNOTS, MAX_COST_W = 12, 24, 12, 12, 64, 16, 16
MAX_SHAPES, MAX_PAIRS, MAX_FREE, MAX_EXTRA_BASES = 64, 128, 4, 3
SHAPE_BOX, SHAPE_SPHERE, SHAPE_DISC = 0, 1, 2
CONTACT_POINT_NORMALS = 1  # Model.contact_flags bit 0
CONTACT_EXPLICIT_LIGHT = 2  # Model.contact_flags bit 1: the explicit law for a light body against a robot link (rounds 1-5)
ABI_VERSION = 9
# error codes of include/mppi_hip.h
MPPI_OK, MPPI_EINVAL, MPPI_EHIP, MPPI_EUNSUPPORTED, MPPI_ESTATE = 0, -1, -2, -3, -4

JOINT_REVOLUTE, JOINT_PRISMATIC = 0, 1
DRIVE_VELOCITY, DRIVE_EFFORT, DRIVE_POSITION = 0, 1, 2
ACTOR_ROBOT, ACTOR_BOX, ACTOR_SPHERE = 0, 1, 2
COST_NONE, COST_POINT_REACH, COST_PANDA_REACH, COST_BOXER_PUSH, COST_PANDA_PICK, COST_PROGRAM = 0, 1, 2, 3, 4, 5
OP_DIST, OP_TILT, OP_YAW_ABS, OP_ALIGN, OP_FORCE_L1, OP_SPEED, OP_DOF_SQ, OP_ABS_DZ, OP_BELOW = 1, 2, 3, 4, 5, 6, 7, 8, 9
SRC_NONE, SRC_RB, SRC_ACTOR, SRC_DOF_XY, SRC_CONST = 0, 1, 2, 3, 4
MAX_TERMS = 16
SAMPLE_HALTON_SPLINE, SAMPLE_EXTERNAL, SAMPLE_NORMAL = 0, 1, 2

_d, _i = C.c_double, C.c_int32


class Body(C.Structure):
    _fields_ = [("parent", _i), ("jtype", _i), ("axis", _d * 3), ("R_tree", _d * 9), ("p_tree", _d * 3),
                ("mass", _d), ("h", _d * 3), ("Io", _d * 6), ("limited", _i), ("pad_", _i),
                ("lower", _d), ("upper", _d), ("effort", _d), ("velocity", _d)]


class Link(C.Structure):
    _fields_ = [("body", _i), ("pad_", _i), ("R", _d * 9), ("p", _d * 3)]


class Actor(C.Structure):
    _fields_ = [("type", _i), ("fixed", _i), ("collision", _i), ("gravity", _i), ("size", _d * 3),
                ("mass", _d), ("friction", _d), ("first_rb", _i), ("n_rb", _i),
                ("noise_sigma_size", _d * 3), ("noise_percentage_mass", _d), ("noise_percentage_friction", _d)]


class Shape(C.Structure):
    _fields_ = [("actor", _i), ("body", _i), ("type", _i), ("rb", _i), ("size", _d * 3), ("R", _d * 9), ("p", _d * 3),
                ("friction", _d)]


class Pair(C.Structure):
    _fields_ = [("a", _i), ("b", _i)]


class Model(C.Structure):
    _fields_ = [("abi_version", _i), ("n_actors", _i), ("actors", Actor * MAX_ACTORS), ("robot_actor", _i),
                ("n_bodies", _i), ("bodies", Body * MAX_BODIES), ("n_links", _i), ("n_rb", _i),
                ("links", Link * MAX_LINKS), ("base_mass", _d), ("base_h", _d * 3), ("base_Io", _d * 6),
                ("drive_mode", _i), ("substeps", _i), ("drive_kd", _d), ("drive_kp", _d), ("dt", _d), ("gravity", _d * 3),
                ("nu", _i), ("cmd_col", (_i * 2) * MAX_BODIES), ("cmd_coef", (_d * 2) * MAX_BODIES),
                ("n_shapes", _i), ("n_pairs", _i), ("shapes", Shape * MAX_SHAPES), ("pairs", Pair * MAX_PAIRS),
                ("ground_friction", _d), ("contact_alpha", _d), ("contact_beta", _d), ("friction_beta", _d), ("contact_ramp_depth", _d),
                ("randomize_seed", _i), ("contact_flags", _i),
                ("n_extra_bases", _i), ("extra_base_actor", _i * MAX_EXTRA_BASES), ("extra_base_mass", _d * MAX_EXTRA_BASES),
                ("extra_base_h", (_d * 3) * MAX_EXTRA_BASES), ("extra_base_Io", (_d * 6) * MAX_EXTRA_BASES)]


class Config(C.Structure):
    _fields_ = [("abi_version", _i), ("num_samples", _i), ("horizon", _i), ("nu", _i), ("k_offset", _i),
                ("k_total", _i), ("sample_null_action", _i), ("use_priors", _i), ("sampling", _i),
                ("n_knots", _i), ("noise_abs_cost", _i), ("want_rollouts", _i), ("viz_link", _i), ("seed", _i),
                ("lambda_", _d), ("rollout_var_discount", _d), ("u_init", _d),
                ("u_min", _d * MAX_NU), ("u_max", _d * MAX_NU), ("noise_sigma_diag", _d * MAX_NU), ("noise_mu", _d * MAX_NU),
                ("spline_basis", _d * (MAX_H * MAX_KNOTS))]


class Term(C.Structure):
    _fields_ = [("op", _i), ("n", _i), ("src", _i * 3), ("idx", _i * 3), ("w", _d), ("p", _d * 8)]


class Cost(C.Structure):
    _fields_ = [("kind", _i), ("link", _i * 4), ("actor", _i * 6), ("w", _d * MAX_COST_W), ("n_terms", _i), ("pad_", _i),
                ("terms", Term * MAX_TERMS)]


_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB_PATH = os.path.join(_PKG_ROOT, "csrc", "libmppi_hip.so")

_fp = C.POINTER(C.c_float)
_vp = C.c_void_p
_SIGNATURES = {
    "mppi_last_error": (C.c_char_p, []),
    "mppi_abi_version": (C.c_int, []),
    "mppi_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mppi_create": (C.c_int, [C.POINTER(Model), C.POINTER(Config), C.c_int, C.POINTER(_vp)]),
    "mppi_jit_info": (C.c_int, [C.c_char_p, C.c_int]),
    "mppi_destroy": (C.c_int, [_vp]),
    "mppi_set_stream": (C.c_int, [_vp, _vp]),
    "mppi_synchronize": (C.c_int, [_vp]),
    "mppi_set_state": (C.c_int, [_vp, _fp, _fp]),
    "mppi_set_state_dev": (C.c_int, [_vp, _vp, _vp]),
    "mppi_get_state": (C.c_int, [_vp, _fp, _fp]),
    "mppi_set_cost": (C.c_int, [_vp, C.POINTER(Cost)]),
    "mppi_set_lambda": (C.c_int, [_vp, C.c_double]),
    "mppi_sample": (C.c_int, [_vp, C.c_uint32]),
    "mppi_sample_normal": (C.c_int, [_vp, C.c_uint32]),
    "mppi_set_noise_dev": (C.c_int, [_vp, _vp]),
    "mppi_set_prior": (C.c_int, [_vp, _fp]),
    "mppi_set_prior_row": (C.c_int, [_vp, C.c_int, _fp]),
    "mppi_set_nominal": (C.c_int, [_vp, _fp]),
    "mppi_get_nominal": (C.c_int, [_vp, _fp]),
    "mppi_set_filter": (C.c_int, [_vp, _fp]),
    "mppi_rollout": (C.c_int, [_vp]),
    "mppi_reduce": (C.c_int, [_vp, _vp]),
    "mppi_record_floats": (C.c_int, [_vp]),
    "mppi_shard_record_count": (C.c_int, [_vp]),
    "mppi_set_record_out": (C.c_int, [_vp, _vp]),
    "mppi_eval_cost": (C.c_int, [_vp, C.c_int, _fp, _fp, _fp, _fp, _fp]),
    "mppi_mailbox_create": (C.c_int, [_vp, C.c_int, C.c_int]),
    "mppi_mailbox_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mppi_mailbox_ptr": (C.c_int, [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "mppi_mailbox_ipc_handle": (C.c_int, [_vp, _vp]),
    "mppi_mailbox_set_peer": (C.c_int, [_vp, C.c_int, _vp]),
    "mppi_mailbox_open": (C.c_int, [_vp, C.c_int, _vp]),
    "mppi_mailbox_gathered": (C.c_int, [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "mppi_exchange": (C.c_int, [_vp]),
    "mppi_exchange_publish": (C.c_int, [_vp]),
    "mppi_exchange_wait": (C.c_int, [_vp]),
    "mppi_exchange_status": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "mppi_exchange_update_step_world": (C.c_int, [_vp, _vp]),
    "mppi_note_graph_update": (C.c_int, [_vp, C.c_int]),
    "mppi_record_dev": (C.c_int, [_vp, C.POINTER(_vp)]),
    "mppi_update": (C.c_int, [_vp, _vp, C.c_int]),
    "mppi_get_action": (C.c_int, [_vp, _fp]),
    "mppi_wait_action": (C.c_int, [_vp, _fp]),
    "mppi_action_dev": (C.c_int, [_vp, C.POINTER(_vp)]),
    "mppi_command": (C.c_int, [_vp, _fp]),
    "mppi_get_costs": (C.c_int, [_vp, _fp]),
    "mppi_get_weights_stats": (C.c_int, [_vp, _fp]),
    "mppi_get_rollouts": (C.c_int, [_vp, _fp]),
    "mppi_get_perturbations": (C.c_int, [_vp, _fp]),
    "mppi_get_noise": (C.c_int, [_vp, _fp]),
    "mppi_sim_reset": (C.c_int, [_vp]),
    "mppi_sim_step": (C.c_int, [_vp, _vp, C.c_int]),
    "mppi_sim_step_horizon": (C.c_int, [_vp, C.c_int]),
    "mppi_sim_step_host": (C.c_int, [_vp, _fp]),
    "mppi_sim_materialise_mirror": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "mppi_mirror_wait": (C.c_int, [_vp, _fp, _fp]),
    "mppi_sim_materialise": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "mppi_sim_accumulate_cost": (C.c_int, [_vp, C.c_int, _vp]),
    "mppi_sim_finish": (C.c_int, [_vp]),
    "mppi_rollout_trajectory": (C.c_int, [_vp]),
    "mppi_materialise_trajectory": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "mppi_reduce_horizon_costs": (C.c_int, [_vp, _vp, _vp]),
    "mppi_materialise_trajectory_link": (C.c_int, [_vp, C.c_int, _vp]),
    "mppi_world_step_from": (C.c_int, [_vp, _vp]),
    "mppi_set_state_from_world": (C.c_int, [_vp, _vp]),
    "mppi_update_step_world": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "mppi_set_profiling": (C.c_int, [_vp, C.c_int]),
    "mppi_kernel_ms": (C.c_int, [_vp, C.c_int, _fp]),
    "mppi_kernel_info": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "mppi_set_wave_clock": (C.c_int, [_vp, C.c_int]),
    "mppi_get_wave_clock": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_int]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class MppiHipError(RuntimeError):
    pass


def load_library(path: str = None) -> C.CDLL:
    """dlopen libmppi_hip.so and attach signatures.  Raises (never falls back) if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MPPI_HIP_LIB") or LIB_PATH  # MPPI_HIP_LIB: a variant build for same-box A/B timing (tools/exp/ab_build.sh)
    if not os.path.exists(p):
        raise MppiHipError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    lib = C.CDLL(p)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype, fn.argtypes = res, args
    if lib.mppi_abi_version() != ABI_VERSION:
        raise MppiHipError(f"ABI mismatch: library {lib.mppi_abi_version()} vs binding {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(lib, rc: int) -> None:
    if rc != 0:
        msg = lib.mppi_last_error()
        raise MppiHipError(f"libmppi_hip error {rc}: {msg.decode() if msg else ''}")


def fptr(arr):
    """float32 numpy array -> float*"""
    return arr.ctypes.data_as(_fp)
