import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "point-cloud-reid_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` via gpurun)")


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: g[k] for k in g.files}
    if "meta" in d:
        d["meta"] = json.loads(str(d["meta"]))
    return d


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    import torch
    return torch.cuda.is_available()


@pytest.fixture(params=["f32", "bf16x3"])
def grad_floor(request):
    """runs a training test once per arithmetic of the gradient products (pcr_amd.train_ops.TRAIN_PRECISION) and yields the
    floor of its gradient yardsticks, relative to a tensor's scale: "f32" (f32-input MFMAs, the reference's arithmetic):
    1e-5, as since round 2; "bf16x3" (the default: dx / dW of the 128 x 128 grouped layers and of the fused attention
    chains as split bf16, ~2^-17 per product, forward values untouched): 3e-5 -- measured worst 1.2e-5."""
    from pcr_amd import train_ops
    prev = train_ops.set_train_precision(request.param)
    yield {"f32": 1e-5, "bf16x3": 3e-5}[request.param]
    train_ops.set_train_precision(prev)
