"""Minimal Hydra/OmegaConf stand-in: enough to accept the override grammar of
bash/run_bimodal_va.sh:22-33 and bash/run_bimodal_at.sh:25-43 of the reference
(`+group=option`, `+group/sub=option`, `a.b.c=value`, `+a.b=value`, `${a.b}` interpolation).

hydra / omegaconf are not installed on the target image; the defaults live in `config_defaults.py`
as plain dicts keyed by (group, option) with the reference's key names (configs/*.yaml there).
"""
from __future__ import annotations

import copy
import re
from typing import Any, Dict, Iterable, List

import yaml


class Config(dict):
    """dict with attribute access (OmegaConf DictConfig look-alike for the keys we use)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_config(obj: Any) -> Any:
    if isinstance(obj, dict):
        return Config({k: to_config(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_config(v) for v in obj]
    return obj


def to_plain(obj: Any) -> Any:
    if isinstance(obj, dict):
        return {k: to_plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [to_plain(v) for v in obj]
    return obj


def _parse_value(text: str) -> Any:
    text = text.strip()
    if text == "":
        return ""
    try:
        val = yaml.safe_load(text)
    except yaml.YAMLError:
        return text
    if isinstance(val, str):        # YAML 1.1 reads `1e-3` / `1e9` as strings; Hydra's grammar reads them as floats
        try:
            return float(val)
        except ValueError:
            return val
    return val


def _set_path(cfg: dict, path: str, value: Any, create: bool):
    keys = path.split(".")
    node = cfg
    for k in keys[:-1]:
        if k not in node or not isinstance(node[k], dict):
            if not create and k not in node:
                raise KeyError(f"override '{path}': no such key '{k}' (prefix with '+' to add)")
            node[k] = Config() if k not in node else node[k]
        node = node[k]
    if not create and keys[-1] not in node:
        raise KeyError(f"override '{path}': no such key (prefix with '+' to add)")
    node[keys[-1]] = to_config(value)


def _lookup(cfg: dict, path: str):
    node = cfg
    for k in path.split("."):
        node = node[k]
    return node


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def resolve(cfg: Config) -> Config:
    """Resolve `${a.b}` references (whole-value references keep their type)."""
    def res(v, depth=0):
        if depth > 20:
            raise ValueError("interpolation loop")
        if isinstance(v, str):
            m = _INTERP.fullmatch(v)
            if m:
                return res(copy.deepcopy(_lookup(cfg, m.group(1))), depth + 1)
            if _INTERP.search(v):
                return _INTERP.sub(lambda mm: str(res(_lookup(cfg, mm.group(1)), depth + 1)), v)
            return v
        if isinstance(v, dict):
            for k in list(v.keys()):
                v[k] = res(v[k], depth)
            return v
        if isinstance(v, list):
            return [res(x, depth) for x in v]
        return v
    return res(cfg)


def compose(overrides: Iterable[str], groups: Dict[str, Dict[str, dict]] = None, root: dict = None) -> Config:
    """Build the run config.  `+model/audio=vit_val` mounts group `model/audio`, option `vit_val` at cfg.model.audio;
    `a.b=c` overrides an existing key; `+a.b=c` adds one."""
    from . import config_defaults as D
    groups = groups if groups is not None else D.GROUPS
    cfg = to_config(copy.deepcopy(root if root is not None else D.ROOT))
    plain: List[tuple] = []
    for ov in overrides:
        ov = ov.strip()
        if not ov:
            continue
        key, _, val = ov.partition("=")
        add = key.startswith("+")
        key = key.lstrip("+")
        if add and key in groups:
            opt = val.strip()
            if opt not in groups[key]:
                raise KeyError(f"config group '{key}' has no option '{opt}' (have {sorted(groups[key])})")
            _set_path(cfg, key.replace("/", "."), copy.deepcopy(groups[key][opt]), create=True)
        else:
            plain.append((key, _parse_value(val), add))
    for key, val, add in plain:       # value overrides apply after all groups are mounted (Hydra semantics)
        _set_path(cfg, key, val, create=add)
    return resolve(cfg)


def to_yaml(cfg: Config) -> str:
    return yaml.safe_dump(to_plain(cfg), sort_keys=False)
