"""Evaluate a checkpoint over a crop directory: the acceptance harness of the north star's "ReID accuracy within
+-0.1 % of the reference checkpoint" (VERDICT r5 next 5).

One call strings together what the reference spreads over its eval hook, mmdet's `multi_gpu_test` and the dataset's
`evaluate`:

    reference                                                               here
    ---------------------------------------------------------------------  ----------------------------------------
    Config.fromfile + build_model (tools/train.py:101-114)                  pcr_amd.config.Config, mmdet3d.models.build_model
    load_checkpoint, mmcv layout {'meta','state_dict'} (`load_from`)        load_checkpoint (strict; `module.` prefix dropped)
    broadcast BN running stats from rank 0 (core/hooks/eval_hook.py:102-108) shard.broadcast_buffers
    val dataset: ReIDDatasetNuscenesFPVal[Even] over the sparse object      loader.CropDirectory + pairs.ObjectTable +
      loader (datasets/reidentification_nuscenes.py:75-249,                   pairs.build_val_pairs + loader.ValPairs
      object_loader_base.py:99-269), `filter_mode`, `min_points`,
      `max_combinations`, `validation_seed`, `subsample_sparse`
    multi_gpu_test: every rank runs forward(return_loss=False) over its     this rank's contiguous shard of the pair list in
      share of the loader, results collected on rank 0 (eval_hook.py:118-)    batches of `val_samples_per_gpu` through
                                                                              ReIDNet.forward_test; shard.gather_logits
    dataset.evaluate: val_match_acc + per-class + the three tables          metrics.evaluate + metrics.evaluate_tables
      (datasets/reidentification_base.py:87-199)

The crops and the `.pth` are licence-gated and absent offline; what CAN be pinned is that this call reproduces, decision
for decision, the accuracy computed from the CPU oracle's logits on the same items (tests/test_gpu_evaluate.py: a toy crop
directory written from tests/golden/pairs_toy.npz, a seeded checkpoint in mmcv's layout, one and two ranks).

Crop directory: `<crop_root>/<token>/<observation>/pts_xyz.bin` (float32 [n,3], object_loader_base.py:247-269).  The
reference keeps class / false-positive flag / visibility of every object in a lamtk pickle that cannot be read without
lamtk; here they come as `meta`: a dict, or a JSON / pickle file (default `<crop_root>/meta.json`),
`{token: {"cls": int | "class_name": str, "fp": bool, "visibility": {observation: level}}}` -- `class_name` goes through
the config's `tracking_classes` and `cls_to_idx`.

Every item draws from numpy's global generator (`subsamplePC`'s resampling, the stand-in dense cloud of a false
positive).  The reference seeds one generator per DataLoader worker, so its items depend on the worker count; here item i
is drawn under `seed_of(seed, i)`, so that the items -- and with them the logits -- do not depend on how many ranks share
the work.
"""
import json
import os
import pickle

import numpy as np
import torch

from . import data as D
from . import loader as LD
from . import metrics, pairs as PR, shard
from .config import Config


def load_checkpoint(model, path, strict=True, map_location="cpu"):
    """mmcv's checkpoint layout ({'meta': .., 'state_dict': .., ['optimizer': ..]}; a bare state_dict is accepted), DDP's
    `module.` prefix dropped, strict by default: a checkpoint that does not fit the configured model is an error, not a
    partially initialised model.  -> the checkpoint's meta dict"""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    sd = ckpt.get("state_dict", ckpt) if isinstance(ckpt, dict) else ckpt
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    model.load_state_dict(sd, strict=strict)
    return ckpt.get("meta", {}) if isinstance(ckpt, dict) else {}


def read_meta(crop_root, meta=None):
    if isinstance(meta, dict):
        return meta
    path = meta or os.path.join(crop_root, "meta.json")
    if str(path).endswith((".pkl", ".pickle")):
        with open(path, "rb") as f:
            return pickle.load(f)
    with open(path) as f:
        return json.load(f)


def seed_of(seed, i):
    """the numpy seed item i is drawn under (see the module docstring)"""
    return (int(seed) * 1000003 + 7919 * int(i) + 17) % (2 ** 32)


def _class_index(entry, val_cfg):
    if "cls" in entry:
        return int(entry["cls"])
    name = entry.get("class_name")
    tracking = val_cfg.get("tracking_classes") or {}
    cls_to_idx = val_cfg.get("cls_to_idx") or {}
    return int(cls_to_idx.get(tracking.get(name, name), -1))


def build_val_set(val_cfg, crop_root, meta=None, max_combinations=None, seed=None, literal_exclusion=False,
                  num_classes=None):
    """the validation dataset the config names, over the crop directory: -> (ValPairs, ObjectTable).
    `type` ending in 'ValEven' -> the even rule (seeds numpy itself), otherwise the FPVal rule under `seed`;
    `sparse_loader.filter_mode` 'pts' or 'pts and vis' (observations WITH a visibility entry only) and
    `sparse_loader.min_points`; `max_combinations` / `validation_seed` / `subsample_sparse` / `subsample_dense` from the
    config unless given."""
    val_cfg = dict(val_cfg or {})
    info = read_meta(crop_root, meta)
    sl = dict(val_cfg.get("sparse_loader") or {})
    mode = sl.get("filter_mode", "pts")
    if mode not in ("pts", "pts and vis"):
        raise NotImplementedError("sparse_loader.filter_mode=%r (the ReID point configs use 'pts' / 'pts and vis')" % mode)
    if num_classes is None:
        num_classes = len(val_cfg["CLASSES"]) if val_cfg.get("CLASSES") else \
            1 + max([_class_index(e, val_cfg) for e in info.values()] + [0])
    crops = LD.CropDirectory(crop_root, load_fraction=float(sl.get("load_fraction", 1.0) or 1.0))
    objs, vis = [], {}
    for tok in sorted(os.listdir(crop_root)):
        d = os.path.join(crop_root, tok)
        if not os.path.isdir(d):
            continue                                        # (meta.json may live inside the crop root)
        e = info[tok]
        vis[tok] = {int(k): v for k, v in (e.get("visibility") or {}).items()}
        frames = {}
        for obs in sorted(os.listdir(d), key=int):
            if mode == "pts and vis" and int(obs) not in vis[tok]:
                continue
            frames[int(obs)] = int(os.stat(os.path.join(d, obs, "pts_xyz.bin")).st_size // 12)
        objs.append(dict(token=tok, cls=_class_index(e, val_cfg), fp=bool(e.get("fp", False)), frames=frames))
    table = PR.ObjectTable(objs, num_classes, min_points=int(sl.get("min_points", 1)))
    seed = int(val_cfg.get("validation_seed", 0) if seed is None else seed)
    mc = int(val_cfg.get("max_combinations", 10) if max_combinations is None else max_combinations)
    even = str(val_cfg.get("type", "ReIDDatasetNuscenesFPValEven")).endswith("ValEven")
    if not even:
        np.random.seed(seed)                                # (the FPVal rule runs on the caller's generator)
    pos, neg = PR.build_val_pairs(table, mc, seed=seed, literal_exclusion=literal_exclusion, even=even)
    nd = int(val_cfg.get("subsample_dense", 0) or 0)
    ds = LD.ValPairs(table, pos, neg, crops.read, int(val_cfg.get("subsample_sparse", 128)), nd,
                     # FakeCompleteLoader (every ReID point config): zeros of shape (3, n) stand in for the aggregated cloud
                     read_dense=(lambda tok: np.zeros((3, max(nd, 1)))), visibility=vis)
    return ds, table


def evaluate_model(model, dataset, batch_size, seed=0, device="cuda", rank=None, world=None, cls_to_idx=None,
                   num_classes=None, on_batch=None):
    """multi_gpu_test + dataset.evaluate for one model over one ValPairs: this rank's contiguous shard of the pairs in
    batches of `batch_size` through forward_test, logits gathered in pair order on every rank, metrics from the gathered
    set.  -> dict(val_match_acc, f1 / precision / recall, per-class and per-bucket accuracies, `tables`, `logits`,
    `targets`, `num_pairs`, `world`)"""
    if rank is None:
        rank, _, world = shard.env_world()
        if not shard.is_dist():
            rank, world = 0, 1
    n = len(dataset)
    lo, hi = shard.shard_range(n, rank, world)
    was_training = model.training
    model.eval()
    shard.broadcast_buffers(model)
    keep = ("val_match_gt", "match_classes", "num_points", "val_vis_gt_all", "is_fp")
    local = []
    with torch.no_grad():
        for b0 in range(lo, hi, batch_size):
            items = []
            for i in range(b0, min(hi, b0 + batch_size)):
                np.random.seed(seed_of(seed, i))
                items.append(dataset[i])
            batch = D.collate_pairs(items, device=device)
            (res,) = model(return_loss=False, **batch)
            if on_batch is not None:
                on_batch(b0, batch, res)
            local.append({k: res[k].detach() for k in ("val_match_preds",) + keep})
    if was_training:
        model.train()
    dev = torch.device(device)

    def gathered(key, width, dtype):
        rows = [r[key].reshape(r[key].shape[0], -1).to(dtype) for r in local]
        mine = torch.cat(rows, 0) if rows else torch.zeros((0, width), dtype=dtype, device=dev)
        if shard.is_dist() and torch.distributed.get_backend() == "gloo":
            mine = mine.cpu()                               # (the CPU rehearsal backend; RCCL gathers device tensors)
        cols = [shard.gather_logits(mine[:, c].contiguous(), n) for c in range(width)]
        return torch.stack(cols, 1)
    # forward_test's per-pair tensors in pair order on every rank (collect_results): float logits / targets, integer
    # classes / sizes / visibility
    full = {"val_match_preds": gathered("val_match_preds", 1, torch.float32)[:, 0],
            "val_match_gt": gathered("val_match_gt", 1, torch.float32)[:, 0],
            "match_classes": gathered("match_classes", 2, torch.float32).long(),
            "num_points": gathered("num_points", 2, torch.float32).long(),
            "val_vis_gt_all": gathered("val_vis_gt_all", 2, torch.float32).long()}
    full = {k: v.cpu() for k, v in full.items()}
    out = metrics.evaluate([full], cls_to_idx={k: v for k, v in (cls_to_idx or {}).items() if v != -1} or None,
                           num_classes=num_classes)
    out["tables"] = metrics.evaluate_tables([full])
    out.update(logits=full["val_match_preds"], targets=full["val_match_gt"], num_pairs=n, world=world)
    return out


def evaluate_checkpoint(cfg_path, ckpt_path, crop_root, max_combinations=None, seed=None, meta=None, batch_size=None,
                        device="cuda", cfg_options=None, literal_exclusion=False, num_classes=None):
    """(config file, mmcv-layout checkpoint, crop directory) -> `val_match_acc` and everything else
    `ReIDDatasetBase.evaluate` logs, on however many ranks torch.distributed is running (one process per GPU; no
    data-path collective -- BatchNorm statistics are broadcast once before, the per-pair results gathered once after)."""
    cfg = Config.fromfile(cfg_path)
    if cfg_options:
        cfg.merge_from_dict(cfg_options)
    from mmdet3d.models import build_model
    import copy
    model = build_model(copy.deepcopy(cfg.model))          # (build_module consumes the `type` keys of the dict it is given)
    ck_meta = load_checkpoint(model, ckpt_path, strict=True)
    model = model.to(device).eval()
    data_cfg = dict(cfg.get("data") or {})
    val_cfg = dict(data_cfg.get("val") or {})
    ds, table = build_val_set(val_cfg, crop_root, meta=meta, max_combinations=max_combinations, seed=seed,
                              literal_exclusion=literal_exclusion, num_classes=num_classes)
    bs = int(batch_size or data_cfg.get("val_samples_per_gpu") or data_cfg.get("samples_per_gpu") or 512)
    seed = int(val_cfg.get("validation_seed", 0) if seed is None else seed)
    out = evaluate_model(model, ds, bs, seed=seed, device=device, cls_to_idx=val_cfg.get("cls_to_idx"),
                         num_classes=table.num_classes)
    out["checkpoint_meta"] = ck_meta
    out["config"] = os.path.abspath(cfg_path)
    guard = getattr(model, "guard_state", lambda: None)()
    if guard is not None:
        out["guard"] = {k: guard[k] for k in ("level", "dlogit", "bound", "sentinel") if k in guard}
    return out


def main(argv=None):
    """python -m pcr_amd.evaluate CONFIG CHECKPOINT CROP_ROOT [--meta FILE] [--max-combinations K] [--seed S]
    (under torch.distributed.run for N GPUs); rank 0 prints the metrics as one JSON line"""
    import argparse
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("config")
    ap.add_argument("checkpoint")
    ap.add_argument("crop_root")
    ap.add_argument("--meta", default=None)
    ap.add_argument("--max-combinations", type=int, default=None)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--batch-size", type=int, default=None)
    args = ap.parse_args(argv)
    rank, local, world = shard.env_world()
    torch.cuda.set_device(local)
    shard.init(device=torch.device("cuda", local))
    out = evaluate_checkpoint(args.config, args.checkpoint, args.crop_root, max_combinations=args.max_combinations,
                              seed=args.seed, meta=args.meta, batch_size=args.batch_size)
    if rank == 0:
        flat = {k: v for k, v in out.items() if isinstance(v, (int, float, str))}
        flat["tables"] = {name: {str(k): v for k, v in metrics.flatten_tables(t).items()}
                          for name, t in out["tables"].items()}
        print(json.dumps(flat))
    if shard.is_dist():
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
