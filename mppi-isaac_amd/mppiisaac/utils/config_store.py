"""Config schema of the examples (reference mppiisaac/utils/config_store.py:9-45).

The reference registers `ExampleConfig` with Hydra's ConfigStore and composes
`defaults: [mppi: X, isaacgym: Y]` from conf/.  Hydra/OmegaConf are optional here: when they are
importable the same registration is done; `load_config` below is a dependency-free composer that
understands exactly the subset the reference's example YAMLs use (a `defaults` list with the groups
`mppi` / `isaacgym`, `base_*` entries meaning "dataclass defaults", and top-level overrides)."""
import os
from dataclasses import dataclass, field, fields
from typing import List, Optional

import yaml

from mppiisaac.planner.isaacgym_wrapper import IsaacGymConfig
from mppiisaac.planner.mppi import MPPIConfig
from mppiisaac.utils.isaacgym_utils import CONF_DIR


@dataclass
class ExampleConfig:
    render: bool = False
    n_steps: int = 1000
    mppi: MPPIConfig = field(default_factory=MPPIConfig)
    isaacgym: IsaacGymConfig = field(default_factory=IsaacGymConfig)
    goal: List[float] = field(default_factory=list)
    nx: int = 0
    actors: List[str] = field(default_factory=list)
    initial_actor_positions: List[List[float]] = field(default_factory=list)


def _load_group(group: str, name: str, cls):
    with open(os.path.join(CONF_DIR, group, f"{name}.yaml")) as f:
        raw = yaml.safe_load(f) or {}
    raw.pop("defaults", None)
    known = {f.name for f in fields(cls)}
    unknown = set(raw) - known
    if unknown:
        raise KeyError(f"conf/{group}/{name}.yaml: unknown keys {sorted(unknown)}")
    return cls(**raw)


def load_config(path_or_dict, overrides: Optional[dict] = None) -> ExampleConfig:
    """Compose an ExampleConfig from an example YAML (path) or an equivalent dict."""
    if isinstance(path_or_dict, dict):
        raw = dict(path_or_dict)
    else:
        with open(path_or_dict) as f:
            raw = yaml.safe_load(f)
    raw.pop("hydra", None)
    groups = {}
    for entry in raw.pop("defaults", []):
        if isinstance(entry, dict):
            groups.update(entry)
    mppi = raw.pop("mppi", None)
    gym = raw.pop("isaacgym", None)
    cfg = ExampleConfig(**raw)
    cfg.mppi = _load_group("mppi", groups["mppi"], MPPIConfig) if "mppi" in groups else MPPIConfig()
    cfg.isaacgym = _load_group("isaacgym", groups["isaacgym"], IsaacGymConfig) if "isaacgym" in groups else IsaacGymConfig()
    for obj, upd in ((cfg.mppi, mppi), (cfg.isaacgym, gym)):
        for k, v in (upd or {}).items():
            if not hasattr(obj, k):
                raise KeyError(k)
            setattr(obj, k, v)
    for k, v in (overrides or {}).items():  # dotted overrides, e.g. {"mppi.num_samples": 4096}
        obj = cfg
        parts = k.split(".")
        for p in parts[:-1]:
            obj = getattr(obj, p)
        if not hasattr(obj, parts[-1]):
            raise KeyError(k)
        setattr(obj, parts[-1], v)
    return cfg


try:  # keep the reference's ConfigStore registration when Hydra is installed
    from hydra.core.config_store import ConfigStore

    cs = ConfigStore.instance()
    for _name in ("config_point_robot", "config_panda", "config_boxer_push", "config_panda_pick"):
        cs.store(name=_name, node=ExampleConfig)
    cs.store(group="mppi", name="base_mppi", node=MPPIConfig)
    cs.store(group="isaacgym", name="base_isaacgym", node=IsaacGymConfig)
except ImportError:
    pass


def load_isaacgym_config(name):
    """reference config_store.py:42-46 composes conf/<name>.yaml through hydra; here the same file goes through
    load_config (hydra / omegaconf are optional in this package)"""
    import os
    conf = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "conf")
    path = os.path.join(conf, name if name.endswith(".yaml") else name + ".yaml")
    return load_config(path)
