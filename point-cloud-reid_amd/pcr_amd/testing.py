"""Deterministic synthetic weights and inputs shared by the golden-vector generator
(oracle/make_golden.py, run once in the dev container against the imported reference),
the parity tests and bench.py.

Weights are *generated*, not stored: a state_dict is a pure function of (manifest, seed),
where the manifest is the reference's parameter/buffer names + shapes (SURVEY.md Appendix A;
committed under tests/golden/*_manifest.json).  That keeps 2.6 MB / 17.6 MB of weights out
of the repository while both sides (reference here, our model everywhere) load bit-identical
tensors.
"""
import json
import zlib

import numpy as np
import torch


def _rng(key, seed):
    return np.random.default_rng([zlib.crc32(key.encode()) & 0xFFFFFFFF, seed])


def seeded_tensor(key, shape, dtype, seed=0):
    """One tensor of the synthetic state_dict, a pure function of (key, shape, seed)."""
    shape = tuple(shape)
    if dtype in ("int64", torch.int64):
        return torch.zeros(shape, dtype=torch.int64)
    g = _rng(key, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "running_mean":
        a = 0.1 * g.standard_normal(shape)
    elif leaf == "running_var":
        a = g.uniform(0.5, 1.5, shape)
    elif len(shape) <= 1 and leaf == "weight":      # BatchNorm / LayerNorm / GroupNorm gain
        a = 1.0 + 0.1 * g.standard_normal(shape)
    elif len(shape) <= 1:                            # any bias
        a = 0.05 * g.standard_normal(shape)
    else:                                            # Linear / Conv weight
        fan_in = int(np.prod(shape[1:]))
        b = np.sqrt(3.0 / fan_in)
        a = g.uniform(-b, b, shape)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def seeded_state_dict(manifest, seed=0):
    """manifest: list of [name, shape, dtype-string] (state_dict order)."""
    return {name: seeded_tensor(name, shape, dtype, seed) for name, shape, dtype in manifest}


def manifest_of(module):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")]
            for k, v in module.state_dict().items()]


def load_manifest(path):
    with open(path) as f:
        return json.load(f)


def synthetic_clouds(n_clouds, n_points, seed, kind="randn"):
    """Object-centred synthetic clouds [n_clouds, n_points, 3] float32.

    randn : unit normal (the survey's probe distribution)
    box   : uniform in a 4 x 2 x 1.5 m box (vehicle-sized crop)
    dup   : box, but half of the points are copies of other points of the same cloud
            (the reference resamples crops *with replacement*, datasets/utils.py:606-621,
            so exact duplicates -- exact distance ties -- are the normal case)
    crop  : what the reference's loader really hands the model: a crop of 32..512 DISTINCT returns on the faces of
            the 4 x 2 x 1.5 m box (a LiDAR sees surfaces), resampled to n_points WITH replacement exactly as
            subsamplePC does (np.random.randint indices) -- every point has ~2..32 exact copies
    """
    g = np.random.default_rng([0x5EED, seed])
    if kind == "randn":
        a = g.standard_normal((n_clouds, n_points, 3))
    elif kind == "crop":
        ext = np.array([4.0, 2.0, 1.5])
        area = np.array([ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[1]])      # faces normal to x, y, z
        a = np.empty((n_clouds, n_points, 3))
        for c in range(n_clouds):
            n_src = int(g.integers(32, 513))
            src = g.uniform(-0.5, 0.5, (n_src, 3)) * ext
            axis = g.choice(3, size=n_src, p=area / area.sum())
            side = g.integers(0, 2, n_src) - 0.5
            src[np.arange(n_src), axis] = side * ext[axis]
            a[c] = src[g.integers(0, n_src, n_points)]
    else:
        a = g.uniform(-0.5, 0.5, (n_clouds, n_points, 3)) * np.array([4.0, 2.0, 1.5])
        if kind == "dup":
            half = n_points // 2
            for c in range(n_clouds):
                src = g.integers(0, half, n_points - half)
                a[c, half:] = a[c, src]
                a[c] = a[c, g.permutation(n_points)]
        elif kind != "box":
            raise ValueError(kind)
    return torch.from_numpy(a.astype(np.float32))


def synthetic_pairs(n_pairs, n_points, seed, kind="randn"):
    c = synthetic_clouds(2 * n_pairs, n_points, seed, kind)
    return c[:n_pairs].contiguous(), c[n_pairs:].contiguous()
