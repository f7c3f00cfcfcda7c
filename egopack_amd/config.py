"""Hydra-compatible configuration without the hydra package (absent on the GPU box).

Implements the subset the reference's entry points use ("Config / flags"):
a root YAML with a ``defaults`` list of config groups (nested groups, ``_self_``, ``override``
entries ignored when they target hydra's own groups), ``key=value`` and ``group/sub=name`` CLI
overrides, ``${path}`` interpolation and ``_target_`` instantiation with positional/keyword extras.
If the real hydra is installed, callers may use it instead; ``instantiate`` accepts its DictConfig.
"""
from __future__ import annotations

import importlib
import re
from pathlib import Path
from typing import Any, Mapping, Sequence

import yaml


class Cfg(dict):
    """dict with attribute access (the part of OmegaConf's DictConfig the entry points rely on)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, Mapping):
        return Cfg({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


_FLOAT = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+$")


def _coerce(node):
    """PyYAML (YAML 1.1) reads ``1e-5`` as a string; OmegaConf reads it as a float.  Follow OmegaConf."""
    if isinstance(node, dict):
        return {k: _coerce(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_coerce(v) for v in node]
    if isinstance(node, str) and _FLOAT.match(node):
        return float(node)
    return node


def _load_yaml(path: Path):
    with open(path) as f:
        return _coerce(yaml.safe_load(f) or {})


def _compose_file(root: Path, rel: str, group_overrides: Mapping[str, str], prefix: str = "") -> dict:
    """Load ``root/rel.yaml`` and merge its defaults list (groups are directories next to it)."""
    doc = _load_yaml(root / f"{rel}.yaml")
    defaults = doc.pop("defaults", [])
    here = str(Path(rel).parent) if "/" in rel else ""
    out: dict = {}
    self_done = False
    for d in defaults:
        if d == "_self_":
            _merge(out, doc)
            self_done = True
            continue
        if isinstance(d, Mapping):
            (g, name), = d.items()
            if g.startswith("override "):
                continue  # hydra/job_logging etc.: logging glue, out of scope
            gpath = f"{prefix}{g}" if prefix else g
            name = group_overrides.get(gpath, name)
            base = f"{here}/{g}" if here else g
            sub = _compose_file(root, f"{base}/{name}", group_overrides, prefix=f"{gpath}/")
            _merge(out, {g: sub})
    if not self_done:
        _merge(out, doc)
    return out


def _merge(dst: dict, src: Mapping):
    for k, v in src.items():
        if isinstance(v, Mapping) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v if not isinstance(v, Mapping) else dict(v)


def _parse_value(s: str):
    try:
        return _coerce(yaml.safe_load(s))
    except yaml.YAMLError:
        return s


def _set_path(cfg: dict, path: str, value):
    keys = path.split(".")
    for k in keys[:-1]:
        cfg = cfg.setdefault(k, {})
    cfg[keys[-1]] = value


_INTERP = re.compile(r"\$\{([^}]+)\}")


def _resolve(node, root):
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node)
        if m:
            return _resolve(_get_path(root, m.group(1)), root)
        return _INTERP.sub(lambda mm: str(_resolve(_get_path(root, mm.group(1)), root)), node)
    return node


def _get_path(cfg, path):
    for k in path.split("."):
        cfg = cfg[k]
    return cfg


def compose(config_dir, config_name: str = "defaults", overrides: Sequence[str] = ()) -> Cfg:
    """hydra.compose equivalent for the reference's config tree."""
    root = Path(config_dir)
    group_over, value_over = {}, []
    for o in overrides:
        k, _, v = o.partition("=")
        k = k.lstrip("+")
        if (root / k).is_dir():  # group selection, e.g. model/temporal_pooling=trn
            group_over[k] = v
        else:
            value_over.append((k, _parse_value(v)))
    cfg = _compose_file(root, config_name, group_over)
    for k, v in value_over:
        _set_path(cfg, k, v)
    return _wrap(_resolve(cfg, cfg))


def instantiate(cfg, *args, **kwargs) -> Any:
    """hydra.utils.instantiate for a mapping with ``_target_`` (non-recursive, as the reference calls
    it with _recursive_=False; nested mappings are passed through as plain config)."""
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    kwargs.pop("_recursive_", None)
    mod, _, name = target.rpartition(".")
    cls = getattr(importlib.import_module(mod), name)
    return cls(*args, **{**cfg, **kwargs})
