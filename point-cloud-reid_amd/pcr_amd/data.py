"""Object-crop data format of the reference, read without mmcv.

On disk (mmdet3d/datasets/object_loader_base.py:247-269): one directory per object and frame,
`<data_root>/<object path>/<frame idx>/pts_<name>.bin` = float32 [n, dim] per feature (`xyz`: dim 3), plus pickle
metadata that carries the paths.  `load_fraction` is the fraction of the points that is LOADED, taken from the end of
the file: the loader seeks to point `n - int(n * load_fraction)` (1.0 = the whole crop).
Crops are then resampled to a fixed size WITH replacement (`subsamplePC`, datasets/utils.py:606-621), which is why
exact duplicate points are the normal case for the kernels.  `collate_pairs` builds the dict the model's
`forward_train` / `forward_test` consume (ReIDNet.py:266-309): lists of per-sample tensors.
"""
import os

import numpy as np
import torch


def load_points(data_root, path, frame_idx, feats=("xyz",), dims=(3,), load_fraction=1.0):
    """-> float32 [n, sum(dims)]: the per-feature files of one object crop, concatenated along the last axis"""
    cols = []
    for name, dim in zip(feats, dims):
        f = os.path.join(data_root, path, str(frame_idx), "pts_%s.bin" % name)
        n = int(os.stat(f).st_size // (4 * dim))
        keep_from = n - int(n * load_fraction)        # the reference's `num_pts -= int(num_pts * load_fraction)`
        cols.append(np.fromfile(f, offset=4 * dim * keep_from, dtype=np.float32).reshape(-1, dim))
    return np.concatenate(cols, axis=-1)


def subsample_pc(pc, n_out, rng=None):
    """reference `subsamplePC`: pc is CHANNEL-major [C, n] (C >= 3); returns [n_out, 3].
    n_out == 0: unchanged (transposed); more than 2 points: n_out indices drawn WITH replacement unless the crop
    already has exactly n_out points; 2 points or fewer: an all-zero cloud."""
    rng = np.random if rng is None else rng
    if n_out == 0:
        return np.moveaxis(pc, 1, 0)
    if pc.shape[1] > 2:
        if pc.shape[0] > 3:
            pc = pc[0:3, :]
        if pc.shape[1] != n_out:
            idx = rng.randint(low=0, high=pc.shape[1], size=n_out, dtype=np.int64)
            pc = pc[:, idx]
        pc = pc.reshape(3, n_out)
    else:
        pc = np.zeros((3, n_out))
    return np.moveaxis(pc, 1, 0)


def collate_pairs(samples, device="cpu"):
    """samples: iterable of dicts with `sparse_1`, `sparse_2` ([N,3] arrays), `id_1`, `id_2` (ints) and optionally
    `dense_*`, `label_*`, `size_*`, `vis_*` -> the model's input dict (lists of per-sample tensors)."""
    out = {}
    for s in samples:
        for k, v in s.items():
            if k.startswith(("sparse_", "dense_")):
                t = torch.as_tensor(np.asarray(v), dtype=torch.float32, device=device)
            else:
                t = torch.as_tensor(np.asarray(v)).reshape(-1).to(device)
            out.setdefault(k, []).append(t)
    for side in ("1", "2"):
        if "dense_" + side not in out and "sparse_" + side in out:
            out["dense_" + side] = out["sparse_" + side]
        if "label_" + side not in out and "sparse_" + side in out:
            out["label_" + side] = [torch.zeros(1, dtype=torch.long, device=device) for _ in out["sparse_" + side]]
    return out
