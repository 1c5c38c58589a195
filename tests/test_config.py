"""CPU: the mmcv-free config loader (pcr_amd/config.py) on its own fixture tree and, when the
reference is present (dev container only), on the reference's configs_reid files, unchanged."""
import os

import pytest

from conftest import GOLDEN
from pcr_amd.config import Config

MINI = os.path.join(GOLDEN, "configs_mini")
REF = "/root/reference/configs_reid"


def test_base_merge_rules():
    cfg = Config.fromfile(os.path.join(MINI, "exp", "sub", "leaf.py"))
    m = cfg.model
    assert m.type == "ReIDNet" and m.eval_only is True
    assert m.backbone_list == [1024, 512, 256]                       # lists replace
    assert m.backbone.nsample == [48, 64, 64] and m.backbone.conv_out == 64   # dicts merge recursively
    assert m.local_stage1 == {"other": 1}                            # _delete_ replaces
    assert len(m.heads) == 1 and m.heads[0].in_features == 8
    assert cfg.data.samples_per_gpu == 256 and cfg.data.val.subsample_sparse == 256
    assert cfg.data.val.path == "root/val"
    assert cfg.seed == 66 and cfg.tags == ["a", "b"] and cfg.hidden == 128
    assert "helper" not in cfg and "os" not in cfg                   # functions / modules are not keys
    assert cfg["model"]["backbone"]["type"] == "Pointnet_Backbone"


def test_duplicate_base_keys_rejected():
    with pytest.raises(KeyError):
        Config.fromfile(os.path.join(MINI, "exp", "clash.py"))


def test_merge_from_dict():
    cfg = Config.fromfile(os.path.join(MINI, "exp", "base_exp.py"))
    cfg.merge_from_dict({"model.backbone.conv_out": 32, "seed": 1})
    assert cfg.model.backbone.conv_out == 32 and cfg.seed == 1 and cfg.model.backbone.type == "Pointnet_Backbone"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the dev container")
def test_reference_configs_load_unchanged_and_build():
    from mmdet3d.models import build_model
    from pcr_amd import testing as T
    cfg = Config.fromfile(os.path.join(
        REF, "reid_nuscenes_pts/num_point_ablation_test/pts_point-transformer_r_nus_det_400e_1024pts.py"))
    assert cfg.model.backbone_list == [1024, 512, 256] and cfg.model.eval_only is True
    assert cfg.data.val.subsample_sparse == 1024 and cfg.data.val_samples_per_gpu == 512
    assert cfg.model.losses_to_use.match is True and cfg.model.losses_to_use.kl is False
    model = build_model(cfg.model)
    assert T.manifest_of(model) == T.load_manifest(os.path.join(GOLDEN, "pt_manifest.json"))
    cfg = Config.fromfile(os.path.join(REF, "reid_nuscenes_pts/testing_pts_pointnet_r_nus_det_500e.py"))
    model = build_model(cfg.model)
    assert T.manifest_of(model) == T.load_manifest(os.path.join(GOLDEN, "pointnet_manifest.json"))
    cfg = Config.fromfile(os.path.join(REF, "reid_waymo_pts/testing_pts_dgcnn_r_waymo_det_400e.py"))
    assert cfg.model.backbone.type == "dgcnn" and cfg.model.use_dgcnn is True
    model = build_model(cfg.model)
    assert T.manifest_of(model) == T.load_manifest(os.path.join(GOLDEN, "dgcnn_manifest.json"))
    cfg = Config.fromfile(os.path.join(
        REF, "reid_waymo_pts/testing_pts_point-transformer_baseline-orig_r_waymo_det_400e.py"))
    assert cfg.model.match_type == "xcorr" and cfg.model.local_stage1.type == "local_self_attention"
    model = build_model(cfg.model)
    assert T.manifest_of(model) == T.load_manifest(os.path.join(GOLDEN, "pt_xcorr_manifest.json"))
    # the wide Point-Transformers (mul = 2 / 4) and the baseline (match_type 'concat', pool_type 'max') configs
    # (the reference's own testing_pts_{1.5M,7M}_* files name a base file that does not exist in its tree --
    # reid_1.5M_point-transformer-point-cat-lin-xcorr_fp.py -- so the reidentifier files they mean are loaded directly)
    for cfg_file, man in (("_base_/reidentifiers/reid_pts_point-transformer-1.5M_point-cat.py", "pt15m"),
                          ("_base_/reidentifiers/reid_pts_point-transformer-7M_point-cat.py", "pt7m")):
        cfg = Config.fromfile(os.path.join(REF, cfg_file))
        with pytest.raises(FileNotFoundError):
            Config.fromfile(os.path.join(REF, "reid_waymo_pts/testing_pts_1.5M_point-transformer_r_waymo_det_800e.py"))
        assert cfg.model.backbone.mul in (2, 4)
        model = build_model(cfg.model)
        assert T.manifest_of(model) == T.load_manifest(os.path.join(GOLDEN, man + "_manifest.json")), cfg_file
    cfg = Config.fromfile(os.path.join(REF, "reid_nuscenes_pts/testing_pts_point-transformer_baseline_r_nus_det_500e.py"))
    assert cfg.model.match_type == "concat" and cfg.model.pool_type == "max"
    model = build_model(cfg.model)              # (incl. the config's 105 M-parameter shape_head, never evaluated)
    man = [m for m in T.manifest_of(model) if not m[0].startswith("shape_head.")]
    assert man == T.load_manifest(os.path.join(GOLDEN, "pt_baseline_manifest.json"))
    del model
    cfg = Config.fromfile(os.path.join(REF, "reid_nuscenes_pts/testing_pts_point-transformer_baseline-stnet_r_nus_det_500e.py"))
    assert cfg.model.match_type == "xcorr-baseline"
    build_model(cfg.model)
    # every point-cloud ReID config of the reference parses
    n, broken = 0, []
    for sub in ("reid_nuscenes_pts", "reid_waymo_pts"):
        for d, _, files in os.walk(os.path.join(REF, sub)):
            for f in files:
                if f.endswith(".py"):
                    try:
                        Config.fromfile(os.path.join(d, f))
                        n += 1
                    except FileNotFoundError:      # a few reference configs name bases that do not exist
                        broken.append(f)
    assert n > 80 and len(broken) <= 10, (n, broken)    # 8 misplaced files in the reference itself


def test_eval_flip_swaps_every_input_of_the_training_path():
    """`eval_flip=True` (ReIDNet.py:144-147 of the reference: 276-279) hands the two sides of every pair to the model in
    swapped order -- clouds, dense clouds, labels and ids together"""
    import copy
    import sys
    import torch
    from conftest import ROOT
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    from mmdet3d.models import build_model
    a = dict(sparse_1=[torch.full((4, 3), float(i)) for i in range(3)], sparse_2=[torch.full((4, 3), 10.0 + i) for i in range(3)],
             dense_1=[torch.full((5, 3), 20.0 + i) for i in range(3)], dense_2=[torch.full((5, 3), 30.0 + i) for i in range(3)],
             label_1=[torch.tensor([i]) for i in range(3)], label_2=[torch.tensor([5 + i]) for i in range(3)],
             id_1=[torch.tensor([40 + i]) for i in range(3)], id_2=[torch.tensor([50 + i]) for i in range(3)])
    outs = {}
    for flip in (False, True):
        cfg = copy.deepcopy(bench.PT_MODEL)
        cfg["eval_flip"] = flip
        outs[flip] = build_model(cfg).preprocess_inputs(**a)
    s1, s2, d1, d2, l1, l2, i1, i2 = outs[False]
    assert s1.shape == (3, 4, 3) and l2.tolist() == [5, 6, 7] and i1.tolist() == [40, 41, 42]
    for x, y in zip(outs[True], (s2, s1, d2, d1, l2, l1, i2, i1)):
        assert torch.equal(x, y)
