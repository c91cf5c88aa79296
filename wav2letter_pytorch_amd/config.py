"""Minimal config objects: attribute + .get access over dicts, lists stay lists
(``cfg.layers[:mid_layers]`` must slice).  Used when Hydra/OmegaConf are absent; an
OmegaConf DictConfig works with the models unchanged."""
from __future__ import annotations

import importlib
import os
import re
from typing import Any


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def to_cfg(o: Any):
    if isinstance(o, dict):
        return Cfg({k: to_cfg(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return [to_cfg(v) for v in o]
    return o


_ALIASES = {
    # reference module paths -> this package (config.yaml:14-16 names decoder.GreedyDecoder)
    'decoder': 'wav2letter_pytorch_amd.decoder',
    'novograd': 'wav2letter_pytorch_amd.novograd',
}


def instantiate(cfg, **kwargs):
    """hydra.utils.instantiate for the ``_target_`` nodes the reference uses
    (base_asr_models.py:22,74,75)."""
    try:
        from hydra.utils import instantiate as hydra_instantiate  # type: ignore
        from omegaconf import DictConfig  # type: ignore
        if isinstance(cfg, DictConfig):
            return hydra_instantiate(cfg, **kwargs)
    except ImportError:
        pass
    args = {k: v for k, v in dict(cfg).items() if k != '_target_'}
    args.update(kwargs)
    mod_name, attr = dict(cfg)['_target_'].rsplit('.', 1)
    try:
        mod = importlib.import_module(mod_name)
    except ImportError:
        if mod_name not in _ALIASES:
            raise
        mod = importlib.import_module(_ALIASES[mod_name])
    return getattr(mod, attr)(**args)


def _interpolate(node, root):
    pat = re.compile(r'\$\{([^}]+)\}')

    def lookup(path):
        cur = root
        for part in path.split('.'):
            cur = cur[part]
        return cur

    if isinstance(node, dict):
        for k, v in list(node.items()):
            node[k] = _interpolate(v, root)
        return node
    if isinstance(node, list):
        return [_interpolate(v, root) for v in node]
    if isinstance(node, str):
        m = pat.fullmatch(node)
        if m:
            return _interpolate(lookup(m.group(1)), root)
        return pat.sub(lambda mm: str(lookup(mm.group(1))), node)
    return node


_FLOAT_RE = re.compile(r'^[-+]?(\d+\.?\d*|\.\d+)([eE][-+]?\d+)?$')


def _yaml_load(text):
    """yaml.safe_load plus OmegaConf's number rule: `1e-5` (no dot) is a float, not a string
    (optimizer/exp_lr_optimizer.yaml:4 relies on it)."""
    import yaml

    def fix(node):
        if isinstance(node, dict):
            return {k: fix(v) for k, v in node.items()}
        if isinstance(node, list):
            return [fix(v) for v in node]
        if isinstance(node, str) and _FLOAT_RE.match(node) and not node.isdigit():
            return float(node)
        return node

    return fix(yaml.safe_load(text))


def load_config(config_dir: str, overrides=()):
    """Hydra-less loader for the reference's config tree (configuration/config.yaml:1-28):
    defaults list, ``# @package model`` groups, ``${a.b}`` interpolation, ``a.b=c`` overrides
    (``model=jasper`` style group overrides select the group file)."""
    with open(os.path.join(config_dir, 'config.yaml')) as f:
        root = _yaml_load(f.read())
    defaults = root.pop('defaults', [])
    groups = {}
    for d in defaults:
        (g, name), = d.items()
        groups[g] = name
    plain = []
    for ov in overrides:
        k, v = ov.split('=', 1)
        if k in groups and '.' not in k:
            groups[k] = v
        else:
            plain.append((k, v))
    merged = {}
    for g, name in groups.items():
        path = os.path.join(config_dir, g, name + '.yaml')
        with open(path) as f:
            text = f.read()
        node = _yaml_load(text) or {}
        m = re.search(r'#\s*@package\s+(\S+)', text)
        pkg = m.group(1) if m else g
        dst = merged.setdefault(pkg, {})
        _deep_update(dst, node)
    _deep_update(merged, root)
    for k, v in plain:
        cur = merged
        parts = k.split('.')
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = _yaml_load(v)
    merged.pop('hydra', None)
    _interpolate(merged, merged)
    return to_cfg(merged)


def _deep_update(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _deep_update(dst[k], v)
        else:
            dst[k] = v
