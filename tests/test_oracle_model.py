"""CPU: the torch-eager restatement (oracle/model_oracle.py) must reproduce the golden vectors
recorded from the imported reference (oracle/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from pcr_amd import testing as T
import model_oracle as MO

TOL = 2e-5   # same ATen ops on the same machine class; observed max |d| is ~1e-6


def _sd(name):
    return T.seeded_state_dict(T.load_manifest(os.path.join(GOLDEN, name + "_manifest.json")), 0)


@pytest.mark.parametrize("case", ["pt_n128_randn", "pt_n128_dup", "pt_n256_box", "pt_n1024_randn"])
def test_pt_oracle_matches_reference_golden(case):
    g = load_golden(case)
    m = g["meta"]
    s1, s2 = T.synthetic_pairs(m["pairs"], m["n"], m["input_seed"], m["kind"])
    st = {}
    with torch.no_grad():
        logits = MO.pt_pairs(_sd("pt"), s1, s2, m["backbone_list"], stages=st)
    assert np.abs(logits.numpy() - g["logits"]).max() < TOL
    for k, v in g.items():
        if k in ("meta", "logits"):
            continue
        got = st[k].numpy()
        if k.endswith("knn_sorted"):
            if m["kind"] == "dup":
                continue          # exact ties: ids interchangeable, covered by feature parity below
            assert (got == v).all(), k
        else:
            assert np.abs(got - v).max() < TOL, (k, np.abs(got - v).max())


@pytest.mark.parametrize("tag,head_ng", [("pt15m", 8), ("pt7m", 16)])
def test_pt_mul_oracle_matches_reference_golden(tag, head_ng):
    """the 1.5M (mul=2) / 7M (mul=4) Point-Transformer configs, recorded from the imported reference"""
    g = load_golden(tag + "_n128_randn")
    m = g["meta"]
    s1, s2 = T.synthetic_pairs(m["pairs"], m["n"], m["input_seed"], m["kind"])
    st = {}
    sd = _sd(tag)
    with torch.no_grad():
        b = s1.shape[0]
        xyz, h = MO.pt_backbone(MO._sub(sd, "backbone."), torch.cat([s1, s2], 0), m["backbone_list"], stages=st)
        st.update(h1=h[:b], h2=h[b:])
        logits = MO.match(sd, h[:b], xyz[:b], h[b:], xyz[b:], st, head_ng=head_ng)
    assert np.abs(logits.numpy() - g["logits"]).max() < TOL
    for k, v in g.items():
        if k in ("meta", "logits"):
            continue
        got = st[k].numpy()
        if k.endswith("knn_sorted"):
            assert (got == v).all(), k
        else:
            assert np.abs(got - v).max() < TOL, (k, np.abs(got - v).max())


def test_pt_baseline_concat_oracle_matches_reference_golden():
    """match_type='concat' + pool_type='max' (reid_pts_point-transformer_baseline.py)"""
    g = load_golden("pt_baseline_n128_randn")
    m = g["meta"]
    s1, s2 = T.synthetic_pairs(m["pairs"], m["n"], m["input_seed"], m["kind"])
    st = {}
    with torch.no_grad():
        logits = MO.pt_pairs_concat(_sd("pt_baseline"), s1, s2, m["backbone_list"], stages=st)
    for k in ("h1", "h2", "pooled1"):
        assert np.abs(st[k].numpy() - g[k]).max() < TOL, k
    assert np.abs(logits.numpy() - g["logits"]).max() < TOL


def test_pointnet_oracle_matches_reference_golden():
    g = load_golden("pointnet_n256_randn")
    m = g["meta"]
    s1, s2 = T.synthetic_pairs(m["pairs"], m["n"], m["input_seed"], m["kind"])
    st = {}
    with torch.no_grad():
        logits = MO.pointnet_pairs(_sd("pointnet"), s1, s2, stages=st)
    for k in ("h1", "h2", "enc_max", "enc_mean"):
        assert np.abs(st[k].numpy() - g[k]).max() < 5e-5, k
    assert np.abs(logits.numpy() - g["logits"]).max() < TOL


@pytest.mark.parametrize("knn", ["pinned_order", "reference_formula"])
def test_dgcnn_oracle_matches_reference_golden(knn):
    """DGCNN: the reference's matmul-based kNN leaves the summation order to the BLAS; the oracle pins one order
    (oracle/pcr_oracle.c:pcr_oracle_knn_feat).  On this fixture both give the reference's neighbour sets exactly."""
    g = load_golden("dgcnn_n256_randn")
    m = g["meta"]
    s1, s2 = T.synthetic_pairs(m["pairs"], m["n"], m["input_seed"], m["kind"])
    st = {}
    with torch.no_grad():
        logits = MO.dgcnn_pairs(_sd("dgcnn"), s1, s2, k=m["k"], stages=st,
                                knn_fn=MO.knn_feat_torch if knn == "reference_formula" else None)
    for i in (1, 2, 3, 4):
        assert (np.sort(st["knn%d" % i].numpy(), -1) == np.sort(g["knn%d" % i].astype(np.int64), -1)).all(), i
    for k in ("h1", "h2", "enc_max", "enc_mean"):
        assert np.abs(st[k].numpy() - g[k]).max() < 5e-5, k
    assert np.abs(logits.numpy() - g["logits"]).max() < TOL


@pytest.mark.parametrize("knn", ["pinned_order", "reference_formula"])
def test_xcorr_local_attention_oracle_matches_reference_golden(knn):
    """match_type='xcorr' (cross -> local_self_attention -> cross -> local), recorded from the imported reference"""
    g = load_golden("pt_xcorr_n128_randn")
    m = g["meta"]
    s1, s2 = T.synthetic_pairs(m["pairs"], m["n"], m["input_seed"], m["kind"])
    st = {}
    with torch.no_grad():
        logits = MO.pt_pairs_xcorr(_sd("pt_xcorr"), s1, s2, m["backbone_list"], knum=m["knum"], stages=st,
                                   knn_fn=MO.knn_feat_torch if knn == "reference_formula" else None)
    for k in ("xc_a", "xc_b"):
        assert np.abs(st[k].numpy() - g[k]).max() < TOL, k
    assert np.abs(logits.numpy() - g["logits"]).max() < TOL


def test_eval_metric_fixture():
    """pcr_amd.metrics against values recorded from the reference's own MatchingEval / accuracy definition"""
    from pcr_amd import metrics
    g = load_golden("eval_metric")
    logits, gt = torch.from_numpy(g["logits"]), torch.from_numpy(g["gt"])
    assert abs(metrics.match_accuracy(logits, gt) - float(g["val_match_acc"])) < 1e-7
    f1 = metrics.f1_precision_recall(metrics.decisions(logits), gt)
    for k, v in f1.items():
        assert abs(v - float(g[k])) < 1e-6, k
    res = [dict(val_match_preds=logits[:40], val_match_gt=gt[:40], num_points=torch.randint(1, 300, (40, 2)), skip=None),
           dict(val_match_preds=logits[40:], val_match_gt=gt[40:], num_points=torch.randint(1, 300, (24, 2)), skip=None)]
    out = metrics.evaluate(res)
    assert abs(out["val_match_acc"] - float(g["val_match_acc"])) < 1e-7 and "val_match_acc_both_ge_1_pts" in out


def test_eval_tables_match_reference_fixture():
    """pcr_amd.metrics.evaluate_points / evaluate_distance / eval_per_visibility: every entry of the three tables
    against the values the imported reference's MatchingEval produced (tests/golden/eval_tables.npz)"""
    import json
    from pcr_amd import metrics
    g = load_golden("eval_tables")
    logits, gt, gt_fp = (torch.from_numpy(g[k]) for k in ("logits", "gt", "gt_fp"))
    got = {"points": metrics.evaluate_points(logits, gt, torch.from_numpy(g["num_points"])),
           "distance": metrics.evaluate_distance(logits, gt, torch.from_numpy(g["dist"])),
           "visibility": metrics.eval_per_visibility(logits, gt_fp, torch.from_numpy(g["vis"]))}
    for name, tables in got.items():
        flat = metrics.flatten_tables(tables)
        keys = json.loads(str(g[name + "_keys"]))
        assert sorted(flat) == keys, name
        want = g[name + "_vals"]
        for k, w in zip(keys, want):
            v = float(flat[k])
            assert (v != v and w != w) or abs(v - w) < 1e-6, (name, k, v, w)
    res = [dict(val_match_preds=logits, val_match_gt=gt, num_points=torch.from_numpy(g["num_points"]),
                val_vis_gt_all=torch.from_numpy(g["vis"]))]
    t = metrics.evaluate_tables(res)
    assert set(t) == {"results_per_points", "results_per_distance", "results_per_visibility"}
