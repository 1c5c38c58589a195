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
