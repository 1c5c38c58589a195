"""`FUSIONMODELS` registry and `build_model`, the model-level plugin API of the reference
(mmdet3d/models/builder.py:5,34-42; used by tools/train.py:114).  mmcv is not required: the
registry below implements the part of mmcv.utils.Registry the ReID configs exercise
(`register_module()` decorator, `build(cfg)` with a `type` key, `get`, `in`)."""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return key in self._module_dict

    def __repr__(self):
        return "Registry(name=%s, items=%s)" % (self._name, sorted(self._module_dict))

    def get(self, key):
        return self._module_dict.get(key)

    def _register(self, cls, name=None, force=False):
        if not inspect.isclass(cls):
            raise TypeError("module must be a class, but got %s" % type(cls))
        name = name or cls.__name__
        if not force and name in self._module_dict:
            raise KeyError("%s is already registered in %s" % (name, self._name))
        self._module_dict[name] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict):
            raise TypeError("cfg must be a dict, but got %s" % type(cfg))
        if "type" not in cfg:
            raise KeyError('`cfg` must contain the key "type", but got %s' % cfg)
        args = dict(cfg)
        kind = args.pop("type")
        if isinstance(kind, str):
            cls = self.get(kind)
            if cls is None:
                raise KeyError("%s is not in the %s registry" % (kind, self._name))
        elif inspect.isclass(kind):
            cls = kind
        else:
            raise TypeError("type must be a str or valid type, but got %s" % type(kind))
        for k, v in (default_args or {}).items():
            args.setdefault(k, v)
        return cls(**args)


FUSIONMODELS = Registry("fusion_models")


def build_fusion_model(cfg, train_cfg=None, test_cfg=None):
    return FUSIONMODELS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_model(cfg, train_cfg=None, test_cfg=None):
    return build_fusion_model(cfg, train_cfg=train_cfg, test_cfg=test_cfg)
