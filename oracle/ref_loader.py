"""TEST INFRASTRUCTURE ONLY -- loads the *reference* (bentherien/point-cloud-reid) in the
development container so golden vectors can be generated from it.

The reference lives at /root/reference (read-only) and never travels to the GPU box, so
nothing under tests/ -m gpu, bench.py or __graft_entry__.smoke() may call this module at
run time.  It is used by oracle/make_golden.py (fixture generator) and by the CPU-only
tests that cross-check our restatement against the real reference when it is present.

The reference's model path (mmdet3d/models/{ReIDNet,backbone_net,pointnet2_utils,attention,
pointnet,lanegcn_nets}.py) is pure PyTorch but imports packages that are absent here
(mmdet, pytorch3d, mmcv) and uses `from fractions import gcd` (removed in py3.9,
lanegcn_nets.py:6).  We install inert stand-ins for those *imports only* -- none of them
is on the arithmetic path we pin (BaseDetector supplies nn.Module plumbing,
chamfer_distance is only used by the disabled shape loss) -- then load the reference
modules by file path under a private package name so they cannot shadow our own
`mmdet3d` mirror package.
"""
import fractions
import importlib.util
import math
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("PCR_REFERENCE_ROOT", "/root/reference")
_PKG = "_pcr_ref"


def available():
    return os.path.isfile(os.path.join(REF_ROOT, "mmdet3d", "models", "ReIDNet.py"))


class _Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, *a, **k):
        def deco(cls):
            self.module_dict[cls.__name__] = cls
            return cls
        return deco


class _BaseDetector(nn.Module):
    """Stand-in for mmdet.models.BaseDetector: only _parse_losses matters
    (mmdet formula: sum of every entry whose key contains 'loss')."""

    def __init__(self, *a, **k):
        super().__init__()

    def _parse_losses(self, losses):
        log_vars = {}
        for k, v in losses.items():
            if isinstance(v, torch.Tensor):
                log_vars[k] = v.mean()
            else:
                log_vars[k] = sum(_v.mean() for _v in v)
        loss = sum(v for k, v in log_vars.items() if "loss" in k)
        log_vars["loss"] = loss
        return loss, {k: float(v) for k, v in log_vars.items()}


_loaded = {}


def _install_stubs():
    if not hasattr(fractions, "gcd"):
        fractions.gcd = math.gcd
    if "mmdet" not in sys.modules:
        mmdet = types.ModuleType("mmdet")
        mmdet_models = types.ModuleType("mmdet.models")
        mmdet_models.BaseDetector = _BaseDetector
        mmdet.models = mmdet_models
        sys.modules["mmdet"] = mmdet
        sys.modules["mmdet.models"] = mmdet_models
    if "pytorch3d" not in sys.modules:
        p3d = types.ModuleType("pytorch3d")
        p3d_loss = types.ModuleType("pytorch3d.loss")

        def chamfer_distance(*a, **k):
            raise RuntimeError("chamfer_distance stub: shape loss is out of scope")

        p3d_loss.chamfer_distance = chamfer_distance
        p3d.loss = p3d_loss
        sys.modules["pytorch3d"] = p3d
        sys.modules["pytorch3d.loss"] = p3d_loss


def _load(name):
    """Load /root/reference/mmdet3d/models/<name>.py as _pcr_ref.<name>."""
    full = f"{_PKG}.{name}"
    if full in sys.modules:
        return sys.modules[full]
    path = os.path.join(REF_ROOT, "mmdet3d", "models", name + ".py")
    spec = importlib.util.spec_from_file_location(full, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    """Returns a namespace with the reference's model-path modules."""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    _install_stubs()
    pkg = types.ModuleType(_PKG)
    pkg.__path__ = []
    sys.modules[_PKG] = pkg
    # ReIDNet.py does `from mmdet3d.models import FUSIONMODELS`; give it a private registry
    # through a temporary sys.modules entry, restored afterwards so our mirror is untouched.
    saved = {k: sys.modules.get(k) for k in ("mmdet3d", "mmdet3d.models")}
    fake = types.ModuleType("mmdet3d")
    fake_models = types.ModuleType("mmdet3d.models")
    fake_models.FUSIONMODELS = _Registry("fusion_models")
    fake.models = fake_models
    sys.modules["mmdet3d"] = fake
    sys.modules["mmdet3d.models"] = fake_models
    try:
        for name in ("pointnet2_utils", "backbone_net", "attention", "pointnet",
                     "lanegcn_nets", "dgcnn_orig", "ReIDNet"):
            _loaded[name] = _load(name)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _loaded["FUSIONMODELS"] = fake_models.FUSIONMODELS
    return types.SimpleNamespace(**_loaded)


def load_ref_config_model(relpath):
    """exec a reference reidentifier config file (plain python, no _base_) -> model dict."""
    path = os.path.join(REF_ROOT, relpath)
    ns = {}
    with open(path) as f:
        exec(compile(f.read(), path, "exec"), ns)
    return ns["model"]


def build_ref_reidnet(relpath="configs_reid/_base_/reidentifiers/reid_pts_point-transformer_point-cat.py",
                      **overrides):
    import copy
    import contextlib
    import io
    ref = load_reference()
    cfg = copy.deepcopy(load_ref_config_model(relpath))
    cfg.update(overrides)
    cfg.pop("type")
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref.ReIDNet.ReIDNet(**cfg)
    return model


def load_dataset_utils():
    """mmdet3d/datasets/utils.py of the reference (MatchingEval, subsamplePC, ...) with inert stand-ins for
    its absent third-party imports (mmcv, torch_cluster); only the pure-torch metric code is exercised."""
    full = _PKG + ".datasets_utils"
    if full in sys.modules:
        return sys.modules[full]

    def mk(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mk("mmcv", Config=object)
    mk("mmcv.runner", get_dist_info=lambda: (0, 1))
    mk("torch_cluster", fps=None, knn=None)
    path = os.path.join(REF_ROOT, "mmdet3d", "datasets", "utils.py")
    spec = importlib.util.spec_from_file_location(full, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    return mod
