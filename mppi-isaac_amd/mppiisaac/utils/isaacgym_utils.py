"""Asset / actor loading (reference mppiisaac/utils/isaacgym_utils.py:14-78).

`load_actor_cfgs` keeps the reference behaviour: conf/actors/<name>.yaml -> ActorWrapper(**yaml)
with PyYAML's SafeLoader (:70-78).  `load_asset` replaces gym.load_asset (:14-29): instead of
importing a URDF into PhysX it returns the compiled model (assets/compiled/*.json) produced by
mppiisaac.backend.urdf_compile from that URDF."""
import glob
import json
import os
from typing import List

import yaml

import mppiisaac
from mppiisaac.planner.isaacgym_wrapper import ActorWrapper

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(mppiisaac.__file__)))
CONF_DIR = os.path.join(PKG_ROOT, "conf")
COMPILED_DIR = os.path.join(PKG_ROOT, "assets", "compiled")


def load_actor_cfgs(actors: List[str]) -> List[ActorWrapper]:
    actor_cfgs = []
    for actor_name in actors:
        # (an actor may also be given as the path of its YAML file: a second robot of the same kind needs a name of its own)
        path = actor_name if str(actor_name).endswith(".yaml") and os.path.exists(actor_name) else os.path.join(CONF_DIR, "actors", f"{actor_name}.yaml")
        with open(path) as f:
            actor_cfgs.append(ActorWrapper(**yaml.load(f, Loader=yaml.SafeLoader)))
    return actor_cfgs


_COMPILED_AT_RUN_TIME = {}   # (urdf path, mtime) -> compiled model


def _find_urdf(urdf_file: str):
    """where gym.load_asset would look (reference isaacgym_utils.py:14-29: asset root <repo>/assets/urdf + urdf_file): an absolute
    path, $MPPI_URDF_ROOT/<urdf_file>, <package>/assets/urdf/<urdf_file>"""
    if not urdf_file:
        return None
    for cand in (urdf_file if os.path.isabs(urdf_file) else None,
                 os.path.join(os.environ["MPPI_URDF_ROOT"], urdf_file) if os.environ.get("MPPI_URDF_ROOT") else None,
                 os.path.join(PKG_ROOT, "assets", "urdf", urdf_file)):
        if cand and os.path.isfile(cand):
            return cand
    return None


def load_asset(actor_cfg: ActorWrapper) -> dict:
    """Model of a robot actor from its `urdf_file` - the counterpart of gym.load_asset (reference isaacgym_utils.py:14-29): the
    compiled fixture under assets/compiled/ when there is one for that URDF, otherwise the URDF itself is compiled HERE, at run
    time (mppiisaac.backend.urdf_compile: tree, frames, inertias from <inertial> or collision hulls, collision primitives).  The
    kernels of a kinematic tree the library has not seen are built on demand by mppi_create (include/mppi_hip.h)."""
    if actor_cfg.type != "robot":
        raise NotImplementedError("only robot actors have a compiled asset; boxes/spheres are described by the actor cfg")
    for path in sorted(glob.glob(os.path.join(COMPILED_DIR, "*.json"))):
        with open(path) as f:
            model = json.load(f)
        if model.get("urdf_file") == actor_cfg.urdf_file:
            return model
    urdf = _find_urdf(actor_cfg.urdf_file)
    if urdf is None:
        raise FileNotFoundError(
            f"urdf_file='{actor_cfg.urdf_file}': no compiled model under {COMPILED_DIR} and no such URDF (absolute path, "
            "$MPPI_URDF_ROOT/<urdf_file>, <package>/assets/urdf/<urdf_file>)")
    key = (urdf, os.path.getmtime(urdf))
    if key not in _COMPILED_AT_RUN_TIME:
        from mppiisaac.backend import capi
        from mppiisaac.backend.urdf_compile import compile_urdf, prune_links
        model = compile_urdf(urdf)
        if len(model["links"]) > capi.MAX_LINKS:   # more URDF links than reported rigid-body rows: keep the observable ones
            model = prune_links(model, keep=[actor_cfg.visualize_link] if actor_cfg.visualize_link else ())
        model["urdf_file"] = actor_cfg.urdf_file
        _COMPILED_AT_RUN_TIME[key] = model
    return _COMPILED_AT_RUN_TIME[key]


def add_ground_plane(gym=None, sim=None) -> None:
    """reference isaacgym_utils.py:61-68 adds a z-up plane with friction 1 to the PhysX scene.  Here the ground is part of
    every contact scene by construction (Scene._contact_scene: GROUND_FRICTION = 1.0); kept as a no-op for callers."""
    return None
