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


def load_asset(actor_cfg: ActorWrapper) -> dict:
    """Compiled model of a robot actor, looked up by its `urdf_file`."""
    if actor_cfg.type != "robot":
        raise NotImplementedError("only robot actors have a compiled asset; boxes/spheres are described by the actor cfg")
    for path in sorted(glob.glob(os.path.join(COMPILED_DIR, "*.json"))):
        with open(path) as f:
            model = json.load(f)
        if model.get("urdf_file") == actor_cfg.urdf_file:
            return model
    raise FileNotFoundError(
        f"no compiled model for urdf_file='{actor_cfg.urdf_file}' under {COMPILED_DIR}; compile it with "
        "mppiisaac.backend.urdf_compile.compile_urdf (tools/compile_models.py) and rebuild the HIP library")


def add_ground_plane(gym=None, sim=None) -> None:
    """reference isaacgym_utils.py:61-68 adds a z-up plane with friction 1 to the PhysX scene.  Here the ground is part of
    every contact scene by construction (Scene._contact_scene: GROUND_FRICTION = 1.0); kept as a no-op for callers."""
    return None
