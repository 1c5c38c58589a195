"""Loader for the reference's python config files (configs_reid/**), reproducing the subset of
mmcv.Config semantics they use, without mmcv:

  * a config file is a python module; every top-level name that is not dunder, a module, a function
    or a class becomes a key;
  * `_base_` is a path or list of paths relative to the including file; bases are loaded first
    (recursively), sibling bases must not define the same top-level key, then the child is merged in;
  * dict values merge recursively into base dicts, everything else (lists included) replaces;
    a child dict carrying `_delete_=True` replaces the base dict instead of merging.

Not supported (unused by configs_reid): `{{ }}` substitutions, custom_imports, json/yaml files.
"""
import copy
import os
import types

BASE_KEY = "_base_"
DELETE_KEY = "_delete_"


class ConfigDict(dict):
    """dict with attribute access, recursively"""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute %r" % name)

    def __setattr__(self, name, value):
        self[name] = value

    def __deepcopy__(self, memo):
        return ConfigDict({copy.deepcopy(k, memo): copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(x):
    if isinstance(x, dict):
        return ConfigDict({k: _wrap(v) for k, v in x.items()})
    if isinstance(x, list):
        return [_wrap(v) for v in x]
    if isinstance(x, tuple):
        return tuple(_wrap(v) for v in x)
    return x


def _merge(child, base):
    """merge dict `child` into a copy of dict `base`"""
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and k in out and isinstance(out[k], dict) and not v.get(DELETE_KEY, False):
            out[k] = _merge(v, out[k])
        else:
            if isinstance(v, dict):
                v = {kk: vv for kk, vv in v.items() if kk != DELETE_KEY}
            out[k] = copy.deepcopy(v)
    return out


def _exec_file(path):
    ns = {"__file__": path, "__name__": "_pcr_config_"}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), ns)
    out = {}
    for k, v in ns.items():
        if k.startswith("__") or isinstance(v, (types.ModuleType, types.FunctionType, type)):
            continue
        out[k] = v
    return out


def _load(path, stack=()):
    path = os.path.abspath(os.path.expanduser(path))
    if not os.path.isfile(path):
        raise FileNotFoundError("config file %s does not exist" % path)
    if not path.endswith(".py"):
        raise IOError("only python configs are supported (got %s)" % path)
    if path in stack:
        raise RecursionError("circular _base_ chain through %s" % path)
    cfg = _exec_file(path)
    bases = cfg.pop(BASE_KEY, None)
    if bases is None:
        return cfg
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        bcfg = _load(os.path.join(os.path.dirname(path), b), stack + (path,))
        dup = set(merged) & set(bcfg)
        if dup:
            raise KeyError("Duplicate key is not allowed among bases: %s (while loading %s)" % (sorted(dup), path))
        merged.update(bcfg)
    return _merge(cfg, merged)


class Config:
    """`Config.fromfile(path)` -> object with attribute / item access, like mmcv.Config"""

    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, "_cfg_dict", _wrap(cfg_dict or {}))
        object.__setattr__(self, "_filename", filename)

    @staticmethod
    def fromfile(filename):
        return Config(_load(filename), filename=os.path.abspath(filename))

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setitem__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))

    def merge_from_dict(self, options):
        """{'a.b.c': v} style overrides (mmcv's --cfg-options)"""
        nested = {}
        for full, v in options.items():
            d = nested
            parts = full.split(".")
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = v
        object.__setattr__(self, "_cfg_dict", _wrap(_merge(nested, dict(self._cfg_dict))))

    def __repr__(self):
        return "Config(path=%s): %r" % (self._filename, dict(self._cfg_dict))
