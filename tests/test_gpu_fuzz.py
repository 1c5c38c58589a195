"""A bounded, deterministic slice of the randomised GPU parity sweeps in tools/ (fixed seeds, fixed case counts,
about half a minute on the GPU box): random shapes through FPS / ball query / both kNNs / local attention / every
dense launch shape against the oracle (indices bit-exact), random grouped SA layers (kNN rows and ragged ball-query
rows, both layouts) and random linear-attention blocks against torch, random training launches against torch
autograd.  The full sweeps stay command-line tools
(`python tools/fuzz_gpu.py 150`), this slice is what the driver's `pytest -m gpu` sees."""
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("seed", [0, 1])
def test_fuzz_point_ops_and_dense_slice(seed):
    import fuzz_gpu
    n, counts = fuzz_gpu.main(budget=120.0, seed=seed, max_cases=120)
    assert n == 120 and len(counts) == 5, counts          # all five case kinds were drawn


def test_fuzz_sa_layers_slice():
    import fuzz_sa
    n, worst = fuzz_sa.main(budget=120.0, seed=3, max_cases=150)
    assert n == 150 and worst < 1e-4


def test_fuzz_attention_blocks_slice():
    import fuzz_attn
    n, worst = fuzz_attn.main(budget=120.0, seed=5, max_cases=150)
    assert n == 150 and worst < 1e-4


def test_fuzz_training_launches_slice():
    """random shapes through the train-dense forward / backward, token norm, attention core, pair pooling and the
    grouped edge MLP against torch autograd (tools/fuzz_train.py: 14 k cases clean in the 15-minute sweep)"""
    import fuzz_train
    n = fuzz_train.main(budget=150.0, seed=11, max_cases=300)
    assert n == 300
