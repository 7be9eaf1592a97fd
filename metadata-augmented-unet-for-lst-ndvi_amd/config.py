"""Minimal stand-in for the reference's global ``CONFIG`` (urban_planner/config.py:43-74).

The reference builds ``CONFIG`` with Hydra's compose API + OmegaConf (struct mode off) and injects
``CONFIG.device`` at run time (src/train.py:99-102).  Hydra/OmegaConf are not needed for the hot
path: this loader reads the same YAML keys into an attribute-style mutable mapping.
"""
from __future__ import annotations

import os

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_YAML = os.path.join(ROOT, "conf", "config.yaml")


class AttrDict(dict):
    """dict with attribute access, nested (like OmegaConf with struct mode off: new keys may be set)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(x):
    if isinstance(x, dict):
        return AttrDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    return x


def load_config(path: str = DEFAULT_YAML) -> AttrDict:
    with open(path) as f:
        cfg = _wrap(yaml.safe_load(f))
    t = cfg.training
    for k in ("learning_rate", "weight_decay", "momentum"):      # "1e-4" parses as a string in YAML 1.1
        if k in t:
            t[k] = float(t[k])
    cfg.MODELS_DIR = os.environ.get("MAU_MODELS_DIR") or os.path.join(ROOT, "models")   # urban_planner/config.py:26-29 analogue
    cfg.device = "cuda:0"                                        # injected by the CLI, as in the reference
    return cfg


CONFIG = load_config(os.environ.get("MAU_CONFIG") or DEFAULT_YAML)
