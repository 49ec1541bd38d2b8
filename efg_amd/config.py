"""YAML -> attribute-dict loader that reads the reference's experiment configs AS THEY ARE.

Counterpart of the slice of efg/config/__init__.py:11-31,34-71 the hot path depends on (omegaconf is not installed, and
a config system is out of scope -- this is the boundary promise "playground/detection.3d configs drop in unchanged"):

* `includes:` -- a list of YAML files merged UNDER the including file (load_yaml, :11-31); like the reference, the
  top-level keys an include brought in (`detection:` of gallary/datasets/waymo.yaml) serve the interpolations and are
  then dropped, and relative paths resolve against the working directory;
* interpolation `${a.b.c}` of other keys (whole value or inside a string, e.g. `${dataset.source.root}${...}.pkl`),
  the resolvers `${oc.env:VAR}` / `${oc.env:VAR,default}` and `${device_count:}` (:66-70);
* the defaults of efg/config/default.yaml:1-66 that this package reads (restated in _DEFAULTS, not loaded from the
  reference tree) merged under the user file (:46);
* `processors:` in the reference's list form (`- Voxelization: {...}` / `- PointShuffle: {p: 1.0}`, $CQ/config.yaml:15-36)
  as well as this repo's older mapping form: a list of single-key mappings becomes a `NamedList`, which still iterates
  like the reference's list and also answers `processors.train.Voxelization` and `"Voxelization" in processors.train`;
* dotted overrides (`a.b.c`, `a.list[2].d`, or through a NamedList by entry name), applied after the merge like the
  reference's `opts` (:72-131), values given as strings decoded with literal_eval (:133-147).
"""
import copy
import os
import re
from ast import literal_eval

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


class NamedList(list):
    """A list of single-key mappings / bare names (the reference's processor lists) that can also be addressed by name:
    `lst.Voxelization`, `lst["Voxelization"]`, `"Voxelization" in lst`, `lst.get("Voxelization")`."""

    def _find(self, name):
        for item in self:
            if isinstance(item, dict) and len(item) == 1 and name in item:
                return item[name]
            if item == name:
                return AttrDict()
        raise KeyError(name)

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        try:
            return self._find(name)
        except KeyError:
            raise AttributeError(name)

    def __getitem__(self, key):
        return self._find(key) if isinstance(key, str) else list.__getitem__(self, key)

    def __contains__(self, key):
        if isinstance(key, str):
            try:
                self._find(key)
                return True
            except KeyError:
                return False
        return list.__contains__(self, key)

    def get(self, name, default=None):
        try:
            return self._find(name)
        except KeyError:
            return default

    def names(self):
        return [next(iter(i)) if isinstance(i, dict) else i for i in self]

    def __deepcopy__(self, memo):
        return NamedList(copy.deepcopy(v, memo) for v in self)


def _is_named(items):
    return len(items) > 0 and all((isinstance(i, dict) and len(i) == 1) or isinstance(i, str) for i in items) \
        and any(isinstance(i, dict) for i in items)


def to_attr(x):
    if isinstance(x, dict):
        return AttrDict({k: to_attr(v) for k, v in x.items()})
    if isinstance(x, (list, tuple)):
        items = [to_attr(v) for v in x]
        return NamedList(items) if _is_named(items) else items
    return x


# efg/config/default.yaml:1-66, the keys this package (or a reference YAML's interpolation) reads
_DEFAULTS = {
    "task": "train",
    "model": {"device": "cuda", "weights": ""},
    "dataloader": {"num_workers": 2, "batch_size": 16},
    "ddp": {"backend": "nccl", "num_gpus": 1, "num_machines": 1, "machine_rank": 0, "find_unused_parameters": False},
    "solver": {"lr_scheduler": {"max_epochs": None, "max_iters": None}, "optimizer": {"lr": None},
               "grad_clipper": {"enabled": False}},
    "trainer": {"type": "DefaultTrainer", "log_interval": 20, "sync_bn": False, "amp": {"enabled": False}},
    "misc": {"debug": False, "seed": -1, "cudnn_benchmark": False},
}

_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _merge(base, over):
    """OmegaConf.merge for plain containers: mappings merge key by key, everything else is replaced."""
    if isinstance(base, dict) and isinstance(over, dict):
        out = dict(base)
        for k, v in over.items():
            out[k] = _merge(base[k], v) if k in base else copy.deepcopy(v)
        return out
    return copy.deepcopy(over)


def _env(arg):
    name, _, default = arg.partition(",")
    name = name.strip()
    if name in os.environ:
        return os.environ[name]
    if _:
        return default.strip()
    raise KeyError("config interpolation ${oc.env:%s}: the environment variable is not set" % name)


def _lookup(expr, root, stack):
    expr = expr.strip()
    if expr.startswith("oc.env:"):
        return _env(expr[len("oc.env:"):])
    if expr.startswith("device_count:"):
        import torch

        return max(1, torch.cuda.device_count())
    if expr in stack:
        raise ValueError("config interpolation cycle through ${%s}" % expr)
    cur = root
    for part in expr.split("."):
        if isinstance(cur, str) and "${" in cur:   # the path runs THROUGH another interpolation (dataset.source.root)
            cur = _resolve(cur, root, stack + (expr,))
        cur = cur[int(part)] if isinstance(cur, list) else cur[part]
    return _resolve(copy.deepcopy(cur), root, stack + (expr,))


def _resolve(node, root, stack=()):
    if isinstance(node, dict):
        return {k: _resolve(v, root, stack) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, stack) for v in node]
    if isinstance(node, str) and "${" in node:
        whole = _INTERP.fullmatch(node)
        if whole:   # the value IS the interpolation: keeps its type (list, mapping, number)
            return _lookup(whole.group(1), root, stack)
        return _INTERP.sub(lambda m: str(_lookup(m.group(1), root, stack)), node)
    return node


def _expand_env_only(text):
    """`includes:` paths are needed before the tree exists: only ${oc.env:...} can appear in them."""
    return _INTERP.sub(lambda m: _env(m.group(1)[len("oc.env:"):]) if m.group(1).startswith("oc.env:") else m.group(0),
                       text)


def _load_raw(path):
    """(the file with its includes merged under it, UNRESOLVED; the top-level keys only the includes define)."""
    with open(path) as f:
        raw = yaml.safe_load(f) or {}
    included = {}
    for inc in raw.pop("includes", None) or []:
        inc = _expand_env_only(inc)
        if not os.path.isabs(inc):
            # the reference joins with "./" (the experiment directory is its cwd); fall back to the including file's folder
            inc = inc if os.path.exists(inc) else os.path.join(os.path.dirname(os.path.abspath(path)), inc)
        included = _merge(included, load_yaml(inc))
    return _merge(included, raw), [k for k in included if k not in raw]


def load_yaml(path, edit=None):
    """One file with its `includes` merged under it and interpolations resolved (efg/config/__init__.py:11-31).
    edit(tree): optional change of the merged tree BEFORE interpolation."""
    merged, include_only = _load_raw(path)
    if edit is not None:
        edit(merged)
    merged = _resolve(merged, merged)
    for key in include_only:   # "known keys to remove": what the include defined only feeds interpolations
        merged.pop(key, None)
    return merged


def _decode(value):
    if not isinstance(value, str):
        return value
    if value == "None":
        return None
    try:
        return literal_eval(value)
    except (ValueError, SyntaxError):
        return value


_FIELD = re.compile(r"^([^\[\]]+)(?:\[(\d+)\])?$")


def _apply_override(root, dotted, value):
    cur = root
    parts = dotted.split(".")
    for i, part in enumerate(parts):
        m = _FIELD.match(part)
        name, index = m.group(1), (int(m.group(2)) if m.group(2) is not None else None)
        last = i == len(parts) - 1
        if isinstance(cur, list):   # a processor list addressed by entry name
            entry = next((it for it in cur if isinstance(it, dict) and len(it) == 1 and name in it), None)
            if entry is None:
                raise AttributeError("override %s: no entry %r in the list" % (dotted, name))
            holder = entry
        else:
            holder = cur
            if name not in holder:
                if index is not None:
                    raise AttributeError("override %s: %r is missing" % (dotted, name))
                holder[name] = {}   # (this repo's overrides may introduce keys; the reference's may not)
        if index is not None:
            if last:
                holder[name][index] = _decode(value)
                return
            cur = holder[name][index]
        elif last:
            holder[name] = _decode(value)
            return
        else:
            if not isinstance(holder[name], (dict, list)):
                holder[name] = {}
            cur = holder[name]


def load_config(path, overrides=None, defaults=True):
    """path: this repo's configs/*.yaml or a reference experiment's config.yaml, unchanged.

    overrides as a dict {dotted: value} (this package's callers): applied to the tree BEFORE interpolation, so
    `${dataset.pc_range}` inside a processor follows an overridden `dataset.pc_range`.
    overrides as the reference's `opts` list (`[k1, v1, k2, v2, ...]` / `["k1=v1", ...]`): applied AFTER the merge and
    the interpolation, values decoded with literal_eval, exactly like Configuration._merge_with_dotlist (:72-147)."""
    if isinstance(overrides, dict):
        early = lambda tree: [_apply_override(tree, k, v) for k, v in overrides.items()]  # noqa: E731
        cfg = load_yaml(path, early)
        overrides = None
    else:
        cfg = load_yaml(path)
    if defaults:
        cfg = _merge(copy.deepcopy(_DEFAULTS), cfg)
    if overrides:
        opts = list(overrides)
        pairs = [o.split("=", 1) for o in opts] if "=" in opts[0] else list(zip(opts[0::2], opts[1::2]))
        for dotted, value in pairs:
            _apply_override(cfg, dotted, value)
    return to_attr(cfg)
