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
