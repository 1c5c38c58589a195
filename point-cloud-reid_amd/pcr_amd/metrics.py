"""Pair-matching evaluation metrics of the ReID datasets, Neptune-free.

Reference: mmdet3d/datasets/reidentification_base.py -- eval_match :69-84, evaluate :87-199 (the
headline `val_match_acc`, :104) and mmdet3d/datasets/utils.py -- MatchingEval.f1_precision_recall
:254-277 (including its operator-precedence quirk: the negative-class recall/precision add 1e-6
AFTER the division).  Inputs are what ReIDNet.forward_test returns, concatenated over batches."""
import torch


def accumulate(results):
    """list of per-batch result dicts (ReIDNet.forward_test) -> dict of concatenated tensors"""
    acc = {}
    for d in results:
        for k, v in d.items():
            if v is not None:
                acc.setdefault(k, []).append(v)
    return {k: torch.cat(v, dim=0) for k, v in acc.items()}


def decisions(logits):
    return (torch.sigmoid(logits) > 0.5).float()


def match_accuracy(logits, targets):
    """val_match_acc = mean((sigmoid(logit) > 0.5) == gt)"""
    return decisions(logits).eq(targets).float().mean().item()


def f1_precision_recall(preds, targets):
    out = {}
    pos = torch.where(targets == 1)[0]
    recall_pos = preds[pos].sum() / (targets[pos].sum() + 1e-6)
    precision_pos = preds[pos].sum() / (preds.sum() + 1e-6)
    f1_pos = 2 * (precision_pos * recall_pos) / (precision_pos + recall_pos + 1e-6)
    out["val_match_f1_pos"] = f1_pos.item()
    out["val_match_recall_pos"] = recall_pos.item()
    out["val_match_precision_pos"] = precision_pos.item()
    neg = torch.where(targets == 0)[0]
    recall_neg = (1 - preds[neg]).sum() / (1 - targets[neg]).sum() + 1e-6
    precision_neg = (1 - preds[neg]).sum() / (1 - preds).sum() + 1e-6
    f1_neg = 2 * (precision_neg * recall_neg) / (precision_neg + recall_neg + 1e-6)
    out["val_match_f1_neg"] = f1_neg.item()
    out["val_match_recall_neg"] = recall_neg.item()
    out["val_match_precision_neg"] = precision_neg.item()
    return out


def per_class_accuracy(logits, targets, match_classes, cls_to_idx, num_classes=None):
    """val_match_acc_<class> over pairs whose first object has that class; `val_match_acc_FP` over pairs
    containing a false-positive detection (class id >= number of real classes)"""
    out = {}
    pred = decisions(logits)
    for name, idx in cls_to_idx.items():
        sel = torch.where(match_classes[:, 0] == idx)[0]
        if len(sel) > 0:
            out["val_match_acc_%s" % name] = pred[sel].eq(targets[sel]).float().mean().item()
    if num_classes is not None:
        sel = torch.where(match_classes.max(1).values >= num_classes)[0]
        if len(sel) > 0:
            out["val_match_acc_FP"] = pred[sel].eq(targets[sel]).float().mean().item()
    return out


def per_point_bucket_accuracy(logits, targets, num_points):
    """accuracy of the pairs whose sparser object has at least 2^i points, for every power of two up to the
    largest cloud (the 'at least both' table of MatchingEval.evaluate_points, utils.py:280-370)"""
    out = {}
    pred = decisions(logits)
    fewest = num_points.min(dim=1).values
    b = 1
    while b <= int(num_points.max().item()):
        sel = torch.where(fewest >= b)[0]
        if len(sel) > 0:
            out["val_match_acc_both_ge_%d_pts" % b] = pred[sel].eq(targets[sel]).float().mean().item()
        b *= 2
    return out


def evaluate(results, cls_to_idx=None, num_classes=None):
    """all of the above from a list of forward_test outputs"""
    r = accumulate(results)
    logits, gt = r["val_match_preds"].float().cpu(), r["val_match_gt"].float().cpu()
    out = {"val_match_acc": match_accuracy(logits, gt)}
    out.update(f1_precision_recall(decisions(logits), gt))
    if cls_to_idx and "match_classes" in r:
        out.update(per_class_accuracy(logits, gt, r["match_classes"].cpu(), cls_to_idx, num_classes))
    if "num_points" in r:
        out.update(per_point_bucket_accuracy(logits, gt, r["num_points"].cpu()))
    return out


def evaluate_tables(results):
    """the reference's results_per_{points,distance,visibility} tables (reidentification_base.py:108-121) from a list
    of forward_test outputs"""
    r = accumulate(results)
    logits, gt = r["val_match_preds"].float().cpu(), r["val_match_gt"].float().cpu()
    out = {}
    if "num_points" in r:
        out["results_per_points"] = evaluate_points(logits, gt, r["num_points"].cpu())
    if "val_vis_gt_all" in r:
        out["results_per_distance"] = evaluate_distance(logits, gt, r["val_vis_gt_all"].cpu())
        out["results_per_visibility"] = eval_per_visibility(logits, gt, r["val_vis_gt_all"].cpu())
    return out


# ---- the reference's per-bucket tables (MatchingEval, datasets/utils.py:280-533) ------------------------------
# Each table entry = f1/precision/recall (above) + accuracy + the positive / negative pair counts of one subset of the
# pairs.  Three subset families per table: 'at_least_one' (the better-observed object of the pair reaches the level),
# 'at_least_both' (both do), 'for_a_pair' (the unordered pair of levels is exactly (i, j)).  NaN entries (empty
# subsets) become -1 where the reference does that.
import itertools

import numpy as np


def _entry(pred, tgt, sel, nan_to_minus1=True):
    p, t = pred[sel], tgt[sel]
    e = f1_precision_recall(p, t)
    e["accuracy"] = (p == t).float().mean().item()
    e["num_observations_pos"] = int((t == 1).sum())
    e["num_observations_neg"] = int((t == 0).sum())
    if nan_to_minus1:
        e = {k: (-1 if isinstance(v, float) and v != v else v) for k, v in e.items()}
    return e


def _in_pair(a, b, lo1, hi1, lo2, hi2):
    first = (lo1 <= a) & (a < hi1) & (lo2 <= b) & (b < hi2)
    second = (lo2 <= a) & (a < hi2) & (lo1 <= b) & (b < hi1)
    return torch.where(first | second)


def evaluate_points(logits, targets, num_points):
    """per power-of-two point-count bucket (utils.py:280-370); num_points (P,2)"""
    pred = decisions(logits)
    a, b = num_points[:, 0], num_points[:, 1]
    edges = [2 ** i for i in range(int(np.log2(num_points.max().item())) + 1)]
    hi, lo = torch.maximum(a, b), torch.minimum(a, b)
    one = {(i, i + 1): _entry(pred, targets, torch.where(edges[i] <= hi)) for i in range(len(edges) - 1)}
    both = {(i, i + 1): _entry(pred, targets, torch.where(edges[i] <= lo)) for i in range(len(edges) - 1)}
    pair = {((i, i + 1), (j, j + 1)): _entry(pred, targets, _in_pair(a, b, edges[i], edges[i + 1], edges[j], edges[j + 1]))
            for i, j in itertools.combinations_with_replacement(range(len(edges) - 1), 2)}
    return dict(at_least_one=one, at_least_both=both, for_a_pair=pair)


def evaluate_distance(logits, targets, dist):
    """per 5-unit distance bucket (utils.py:372-460); dist (P,2).  'at_least_one': the NEARER object is within the
    bucket's lower edge, 'at_least_both': the farther one is"""
    pred = decisions(logits)
    a, b = dist[:, 0], dist[:, 1]
    edges = [5 * i for i in range(int(dist.max().item() / 5) + 3)]
    near, far = torch.minimum(a, b), torch.maximum(a, b)
    one = {(i, i + 1): _entry(pred, targets, torch.where(near <= edges[i])) for i in range(len(edges) - 1)}
    both = {(i, i + 1): _entry(pred, targets, torch.where(far <= edges[i])) for i in range(len(edges) - 1)}
    pair = {((i, i + 1), (j, j + 1)): _entry(pred, targets, _in_pair(a, b, edges[i], edges[i + 1], edges[j], edges[j + 1]))
            for i, j in itertools.combinations_with_replacement(range(len(edges) - 1), 2)}
    return dict(at_least_one=one, at_least_both=both, for_a_pair=pair)


def eval_per_visibility(logits, targets, vis_classes, levels=(0, 1, 2, 3)):
    """per visibility level (utils.py:463-533); vis_classes (P,2) (or (P,2,1)); pairs with target -1 (false-positive
    detections) are left out.  Only the 'for_a_pair' entries have their NaNs replaced, as in the reference."""
    pred = decisions(logits)
    keep = targets != -1
    pred, tgt, vis = pred[keep], targets[keep], vis_classes[keep]
    if vis.dim() == 3:
        vis = vis.squeeze(2)
    a, b = vis[:, 0], vis[:, 1]
    hi, lo = torch.maximum(a, b), torch.minimum(a, b)
    one = {x: _entry(pred, tgt, torch.where(hi >= x), nan_to_minus1=False) for x in levels}
    both = {x: _entry(pred, tgt, torch.where(lo >= x), nan_to_minus1=False) for x in levels}
    pair = {(x, y): _entry(pred, tgt, torch.where(((a == x) & (b == y)) | ((a == y) & (b == x))))
            for x, y in itertools.combinations_with_replacement(levels, 2)}
    return dict(at_least_one=one, at_least_both=both, for_a_pair=pair)


def flatten_tables(tables):
    """nested table dict -> {'family/key/metric': value} (what the reference's make_tup_str + json.dump store)"""
    out = {}
    for fam, rows in tables.items():
        for key, entry in rows.items():
            for m, v in entry.items():
                out["%s/%s/%s" % (fam, key, m)] = v
    return out
