"""GPU: the acceptance harness end to end (pcr_amd/evaluate.py: config file -> build_model -> mmcv-layout checkpoint ->
BN broadcast -> ValPairs over this rank's shard -> forward_test -> gather -> metrics) on a toy crop directory written from
tests/golden/pairs_toy.npz with a seeded checkpoint: `val_match_acc` must equal the accuracy computed from the CPU
oracle's logits on the same items, decision for decision, on one rank and on two (gloo, both on cuda:0).  No reference
Python is involved: the config below is this build's own file, the crops are regenerated from seeds.
(Reference: mmdet3d/core/hooks/eval_hook.py:102-128, mmdet3d/datasets/reidentification_base.py:87-199.)"""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest
import torch

from conftest import ROOT
from pcr_amd import evaluate as EV
from pcr_amd import metrics
from pcr_amd import testing as T
from test_evaluate import VAL_CFG, write_toy_crops

pytestmark = pytest.mark.gpu

CONFIG = textwrap.dedent("""
    # own test fixture (not a reference file): a Point-Transformer ReIDNet in the layout of configs_reid/*
    _base_ = ["./data.py"]
    model = dict(
        type="ReIDNet", hidden_size=128, combine="point-cat", match_type="xcorr_eff", pool_type="both",
        backbone_list=[128, 64, 32], output_sequence_size=64, eval_only=True,
        backbone=dict(type="Pointnet_Backbone", input_channels=0, use_xyz=True, conv_out=64),
        match_head=[dict(type="LinearRes", n_in=128, n_out=128, norm="GN", ng=8),
                    dict(type="Linear", in_features=128, out_features=1)],
        downsample=None, cls_head=None, fp_head=None, shape_head=None,
        cross_stage1=dict(type="corss_attention", d_model=64, nhead=2, attention="linear"),
        cross_stage2=dict(type="corss_attention", d_model=64, nhead=2, attention="linear"),
        local_stage1=dict(), local_stage2=dict(),
        losses_to_use=dict(kl=False, match=True, cls=False, shape=False, fp=False, triplet=False))
""")
DATA = "data = dict(samples_per_gpu=8, val_samples_per_gpu=24, val=%r)\n"


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    import bench
    d = tmp_path_factory.mktemp("eval")
    crops = str(d / "crops")
    os.makedirs(crops)
    write_toy_crops(crops)
    val = dict(VAL_CFG, subsample_sparse=128)
    (d / "data.py").write_text(DATA % (val,))
    (d / "exp.py").write_text(CONFIG)
    man = T.load_manifest(os.path.join(ROOT, "tests", "golden", "pt_manifest.json"))
    sd = T.seeded_state_dict(man, 0)
    # seeded weights put every logit on one side of the threshold (all "no match": accuracy 0.5 whatever the model does);
    # move the head's bias into the widest gap near the median of the oracle's logits so that about half of the decisions
    # flip and "decision for decision" means something
    want, _ = oracle_logits(sd, val, crops)
    srt = torch.sort(want).values
    lo = len(srt) // 2 - 10
    gaps = srt[lo + 1:lo + 21] - srt[lo:lo + 20]
    j = int(gaps.argmax())
    sd["match_head.1.bias"] = sd["match_head.1.bias"] - 0.5 * (srt[lo + j] + srt[lo + j + 1])
    ckpt = str(d / "epoch_500.pth")
    torch.save({"meta": {"epoch": 500, "iter": 12345}, "state_dict": {"module." + k: v for k, v in sd.items()}}, ckpt)
    return str(d / "exp.py"), ckpt, crops, sd, val


def oracle_logits(sd, val, crops):
    """the CPU oracle on the SAME items: the harness's dataset and per-item seeds, rebuilt independently of its loop"""
    import model_oracle as MO
    ds, _ = EV.build_val_set(val, crops)
    s1, s2, gt = [], [], []
    for i in range(len(ds)):
        np.random.seed(EV.seed_of(val["validation_seed"], i))
        it = ds[i]
        s1.append(torch.as_tensor(np.asarray(it["sparse_1"]), dtype=torch.float32))
        s2.append(torch.as_tensor(np.asarray(it["sparse_2"]), dtype=torch.float32))
        gt.append(float(it["id_1"] == it["id_2"]))
    with torch.no_grad():
        logits = MO.pt_pairs(sd, torch.stack(s1), torch.stack(s2), [128, 64, 32])
    return logits, torch.tensor(gt)


def test_one_rank_reproduces_the_oracle_accuracy_decision_for_decision(setup):
    cfg, ckpt, crops, sd, val = setup
    out = EV.evaluate_checkpoint(cfg, ckpt, crops)
    want, gt = oracle_logits(sd, val, crops)
    assert out["num_pairs"] == len(want) == 2 * int(gt.sum()) and out["world"] == 1
    assert out["checkpoint_meta"] == {"epoch": 500, "iter": 12345}
    assert torch.equal(out["targets"], gt)
    err = float((out["logits"] - want).abs().max())
    margin = float(want.abs().min())
    print(json.dumps(dict(pairs=len(want), max_abs_dlogit_vs_oracle=err, smallest_abs_logit=margin,
                          val_match_acc=out["val_match_acc"], guard=out.get("guard", {}).get("level"))))
    assert err < 1e-4, err
    assert margin > 2e-4                                              # (no decision sits on the threshold)
    dec = metrics.decisions(want)
    assert 0.3 < float(dec.mean()) < 0.7                              # (both decisions occur: the fixture's bias shift)
    assert torch.equal(metrics.decisions(out["logits"]), metrics.decisions(want))
    assert out["val_match_acc"] == metrics.match_accuracy(want, gt)
    ref = metrics.evaluate([dict(val_match_preds=want, val_match_gt=gt)])
    for k in ("val_match_f1_pos", "val_match_recall_pos", "val_match_precision_pos", "val_match_f1_neg"):
        assert out[k] == ref[k], k
    assert {"results_per_points", "results_per_distance", "results_per_visibility"} <= set(out["tables"])
    assert "val_match_acc_car" in out and "val_match_acc_pedestrian" in out and "val_match_acc_FP" in out
    # max_combinations / seed overrides reach the pair set
    small = EV.evaluate_checkpoint(cfg, ckpt, crops, max_combinations=1)
    assert small["num_pairs"] < out["num_pairs"]
    # a checkpoint that does not fit the configured model is an error
    bad = dict(torch.load(ckpt, weights_only=False))
    bad["state_dict"] = {k: v for k, v in bad["state_dict"].items() if "match_head" not in k}
    torch.save(bad, ckpt + ".bad")
    with pytest.raises(RuntimeError):
        EV.evaluate_checkpoint(cfg, ckpt + ".bad", crops)


WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, os.path.join(%(root)r, "point-cloud-reid_amd"))
    import torch
    from pcr_amd import evaluate as EV, shard
    rank, local, world = shard.init(backend="gloo")
    torch.cuda.set_device(0)
    if rank == 1:
        # the other rank's BatchNorm statistics must not matter: rank 0's are broadcast before the evaluation
        _load = EV.load_checkpoint
        def drifted(model, path, **kw):
            meta = _load(model, path, **kw)
            for n, b in model.named_buffers():
                if n.endswith("running_mean"):
                    b.add_(0.5)
            return meta
        EV.load_checkpoint = drifted
    out = EV.evaluate_checkpoint(%(cfg)r, %(ckpt)r, %(crops)r)
    if rank == 0:
        json.dump(dict(acc=out["val_match_acc"], logits=out["logits"].tolist(), world=out["world"],
                       pairs=out["num_pairs"]), open(%(out)r, "w"))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_ranks_reproduce_the_one_rank_evaluation(setup, tmp_path):
    cfg, ckpt, crops, sd, val = setup
    one = EV.evaluate_checkpoint(cfg, ckpt, crops)
    script, outp = tmp_path / "w.py", tmp_path / "out.json"
    script.write_text(WORKER % dict(root=ROOT, cfg=cfg, ckpt=ckpt, crops=crops, out=str(outp)))
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    two = json.load(open(outp))
    assert two["world"] == 2 and two["pairs"] == one["num_pairs"]
    got = torch.tensor(two["logits"])
    assert float((got - one["logits"]).abs().max()) < 1e-4
    assert torch.equal(metrics.decisions(got), metrics.decisions(one["logits"])) and two["acc"] == one["val_match_acc"]
