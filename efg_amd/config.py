"""Tiny YAML -> attribute-dict loader for the keys the hot path reads (the reference's OmegaConf
loader, efg/config/__init__.py:34-162, is out of scope; omegaconf is not installed).  Supports
`${a.b.c}` interpolation of other keys, which the ConQueR YAML uses (config.yaml:45,82-85,117-123)."""
import copy
import re

import yaml


class AttrDict(dict):
    """dict with attribute access; missing attributes raise AttributeError (so deepcopy works)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})

    def pop(self, k, *a):  # noqa: D401
        return dict.pop(self, k, *a)


def to_attr(x):
    if isinstance(x, dict):
        return AttrDict({k: to_attr(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        return [to_attr(v) for v in x]
    return x


_REF = re.compile(r"^\$\{([A-Za-z0-9_.]+)\}$")


def _resolve(node, root):
    if isinstance(node, dict):
        return {k: _resolve(v, root) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = _REF.match(node)
        if m:
            cur = root
            for part in m.group(1).split("."):
                cur = cur[part]
            return _resolve(copy.deepcopy(cur), root)
    return node


def load_config(path, overrides=None):
    with open(path) as f:
        raw = yaml.safe_load(f)
    for dotted, value in (overrides or {}).items():
        cur = raw
        parts = dotted.split(".")
        for p in parts[:-1]:
            cur = cur.setdefault(p, {})
        cur[parts[-1]] = value
    return to_attr(_resolve(raw, raw))
