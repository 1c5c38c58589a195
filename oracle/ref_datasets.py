"""TEST INFRASTRUCTURE ONLY -- imports the *reference's* ReID dataset classes in the development container so that the
pair rules of pcr_amd/loader.py (training) and pcr_amd/pairs.py (validation) can be pinned to fixtures recorded from
them (oracle/make_golden.py gen_pairs -> tests/golden/pairs_*.npz).  Never imported by the product, bench.py or the
-m gpu tests; /root/reference does not travel.

Loaded from /root/reference/mmdet3d/datasets, unmodified, under a private package name:
    utils.py (subsamplePC, set_seeds, to_tensor, get_or_create_nuscenes_dict), object_loader_base.py
    (ObjectLoaderSparseBase: filtered observation lists, power-of-two buckets, get_random_frame[_even],
    get_class_list_density, load_points; FakeCompleteLoader), reidentification_base.py (ReIDDatasetBase),
    reidentification_nuscenes.py (ReIDDatasetNuscenesFP, ...FPVal, ...FPValEven).

Stand-ins for ABSENT THIRD-PARTY packages only (none of them is on the sampling path that is pinned):
    mmdet.datasets.DATASETS / mmcv.utils.Registry    -- a name -> class table
    <datasets>.builder.build_dataset                  -- the registry lookup mmcv's build_from_cfg performs (the
                                                        reference's builder.py itself imports mmdet / mmcv wrappers)
    mmcv.parallel.DataContainer                       -- holds `.data`
    mmcv.runner.get_dist_info, mmcv.is_str, torch_cluster.{fps,knn}
    lamtk.aggregation.loader.Loader                   -- the metadata holder ObjectLoaderSparseBase derives from: keeps
                                                        obj_infos / scene_infos / frame_infos, data_root, load_feats,
                                                        load_dims, load_fraction (the attributes the reference's own
                                                        load_points reads, object_loader_base.py:247-269)
    lamtk.aggregation.utils.{filter_metadata_by_scene_ids, combine_metadata}  -- unused identities
The nuScenes token -> id pickle the dataset's constructor reads (`data/lstk/instance_token_to_id.pkl`, made from the
licence-gated dataset by get_or_create_nuscenes_dict) is written into a scratch directory from the toy table and read
back by the REFERENCE's own function.
"""
import importlib.util
import os
import pickle
import sys
import types

import numpy as np

import ref_loader

REF_DS = os.path.join(ref_loader.REF_ROOT, "mmdet3d", "datasets")
_PKG = "_pcr_ref_ds"
_loaded = {}


class _Registry:
    def __init__(self, name="registry"):
        self.name, self.module_dict = name, {}

    def register_module(self, *a, **k):
        def deco(cls):
            self.module_dict[cls.__name__] = cls
            return cls
        return deco

    def get(self, name):
        return self.module_dict[name]


class _DataContainer:
    def __init__(self, data, *a, **k):
        self.data = data


class _Loader:
    """stand-in for lamtk.aggregation.loader.Loader (see the module docstring)"""

    def __init__(self, metadata=None, data_root="", load_feats=("xyz",), load_dims=(3,), load_fraction=1.0, **kwargs):
        metadata = metadata or {}
        self.obj_infos = metadata.get("obj_infos", {})
        self.scene_infos = metadata.get("scene_infos", {})
        self.frame_infos = metadata.get("frame_infos", {})
        self.data_root, self.load_feats, self.load_dims = data_root, list(load_feats), list(load_dims)
        self.load_fraction = load_fraction


def _mk(name, **attrs):
    if name in sys.modules:
        m = sys.modules[name]
    else:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    for k, v in attrs.items():
        if not hasattr(m, k):
            setattr(m, k, v)
    return m


def available():
    return os.path.isfile(os.path.join(REF_DS, "reidentification_nuscenes.py"))


def load():
    """-> namespace(utils, object_loader_base, reidentification_base, reidentification_nuscenes, DATASETS)"""
    if _loaded:
        return types.SimpleNamespace(**_loaded)
    if not available():
        raise RuntimeError("reference datasets not present at %s" % REF_DS)
    utils = ref_loader.load_dataset_utils()          # the reference's datasets/utils.py (mmcv / torch_cluster stubs)
    registry = _Registry("dataset")
    mmcv = _mk("mmcv", is_str=lambda x: isinstance(x, str))
    if not hasattr(mmcv, "is_str"):
        mmcv.is_str = lambda x: isinstance(x, str)
    _mk("mmcv.runner", get_dist_info=lambda: (0, 1))
    _mk("mmcv.parallel", DataContainer=_DataContainer)
    _mk("mmcv.utils", Registry=_Registry)
    _mk("mmdet")
    _mk("mmdet.datasets", DATASETS=registry)
    registry = sys.modules["mmdet.datasets"].DATASETS
    _mk("lamtk")
    _mk("lamtk.aggregation")
    _mk("lamtk.aggregation.loader", Loader=_Loader)
    _mk("lamtk.aggregation.utils", filter_metadata_by_scene_ids=lambda md, ids: md, combine_metadata=lambda mds: mds[0])

    pkg = types.ModuleType(_PKG)
    pkg.__path__ = []
    sys.modules[_PKG] = pkg
    sys.modules[_PKG + ".utils"] = utils
    builder = types.ModuleType(_PKG + ".builder")

    def build_dataset(cfg, default_args=None):
        cfg = dict(cfg)
        return registry.get(cfg.pop("type"))(**cfg)
    builder.build_dataset = build_dataset
    sys.modules[_PKG + ".builder"] = builder

    # `from mmdet3d.datasets.utils import ...` inside the reference files: the reference's utils, for the import only
    saved = {k: sys.modules.get(k) for k in ("mmdet3d", "mmdet3d.datasets", "mmdet3d.datasets.utils")}
    fake = types.ModuleType("mmdet3d")
    fake.__path__ = []
    fake_ds = types.ModuleType("mmdet3d.datasets")
    fake_ds.__path__ = []
    sys.modules.update({"mmdet3d": fake, "mmdet3d.datasets": fake_ds, "mmdet3d.datasets.utils": utils})
    try:
        for name in ("object_loader_base", "reidentification_base", "reidentification_nuscenes"):
            full = "%s.%s" % (_PKG, name)
            spec = importlib.util.spec_from_file_location(full, os.path.join(REF_DS, name + ".py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[full] = mod
            spec.loader.exec_module(mod)
            _loaded[name] = mod
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _loaded["utils"] = utils
    _loaded["DATASETS"] = registry
    return types.SimpleNamespace(**_loaded)


# ---------------------------------------------------------------------------------------------- the toy world --
CLASSES = ["car", "pedestrian"]
TRACKING = {"vehicle.car": "car", "human.pedestrian.adult": "pedestrian"}


def toy_objects(seed=0, n_true=10, n_fp=6):
    """a small object table: tokens in sorted order, class, observation number -> number of points, visibility.
    Point counts span several power-of-two buckets; some objects have fewer than three observations (dropped by the
    reference's `temp > 2`), one is of an untracked class."""
    g = np.random.RandomState(seed)
    objs = []
    names = list(TRACKING)
    for i in range(n_true):
        nobs = int(g.randint(2, 7)) if i not in (3,) else 2
        cls = names[i % 2] if i != 7 else "movable_object.barrier"
        nums = sorted(g.choice(np.arange(40), nobs, replace=False).tolist())
        objs.append(dict(token="obj%02d" % i, class_name=cls, fp=False,
                         frames={int(n): int(2 ** g.uniform(2.0, 8.5)) for n in nums},
                         visibility={int(n): int(g.randint(1, 5)) for n in nums}))
    for i in range(n_fp):
        nobs = int(g.randint(1, 5))
        nums = sorted(g.choice(np.arange(40), nobs, replace=False).tolist())
        objs.append(dict(token="FP_%02d" % i, class_name=names[i % 2], fp=True,
                         frames={int(n): int(2 ** g.uniform(2.0, 8.5)) for n in nums},
                         visibility={int(n): int(g.randint(1, 5)) for n in nums}))
    objs.sort(key=lambda o: o["token"])       # ('FP_..' < 'obj..': os.listdir order of pcr_amd.loader.CropDirectory)
    return objs


def crop_points(token, obs, npts):
    """the crop of (token, observation): seeded by both, so that generator and test rebuild the same directory"""
    h = (sum(ord(c) * (i + 1) for i, c in enumerate(token)) * 1009 + int(obs) * 9176 + 12345) % (2 ** 31 - 1)
    return np.random.RandomState(h).randn(int(npts), 3).astype(np.float32)


def write_crops(root, objs):
    for o in objs:
        for n, npts in o["frames"].items():
            d = os.path.join(root, o["token"], str(n))
            os.makedirs(d, exist_ok=True)
            crop_points(o["token"], n, npts).tofile(os.path.join(d, "pts_xyz.bin"))


def build_reference_dataset(kind, objs, root, scratch, subsample_sparse, subsample_dense, seed, max_combinations=3):
    """kind 'train' -> ReIDDatasetNuscenesFP, 'val' -> ReIDDatasetNuscenesFPVal, 'val_even' -> ...FPValEven, over the toy
    table with the crops under `root`; numpy's global generator is seeded with `seed` right before construction (the
    reference's train script calls set_seeds first; the ValEven class re-seeds itself with validation_seed)."""
    R = load()
    olb = R.object_loader_base
    reg = R.DATASETS

    if "ToySparseLoader" not in reg.module_dict:
        @reg.register_module()
        class ToySparseLoader(olb.ObjectLoaderSparseBase):
            """ObjectLoaderSparseWaymo.__init__ (object_loader_base.py:357-372) without the metadata files"""

            def __init__(self, metadata, **kwargs):
                super().__init__(metadata=metadata, **kwargs)
                self.obj_id_to_nums = self.collect_obj_id_to_nums(self.min_points)
                self.get_buckets(np.arange(0, len(self.obj_id_to_nums)))
                self.get_all_buckets(np.arange(0, len(self.obj_id_to_nums)))

            def load(self, *a, **k):
                return self.load_points(*a, **k)

    obj_infos = {o["token"]: dict(id=o["token"], class_name=o["class_name"], path=o["token"],
                                  num_pts=dict(o["frames"]), visibility=dict(o["visibility"])) for o in objs}
    os.makedirs(os.path.join(scratch, "data", "lstk"), exist_ok=True)
    with open(os.path.join(scratch, "data", "lstk", "instance_token_to_id.pkl"), "wb") as f:
        pickle.dump({o["token"]: i for i, o in enumerate(objs)}, f)
    cls_to_idx = {c: i for i, c in enumerate(CLASSES)}
    cls_to_idx["none_key"] = -1
    cls_to_idx_fp = {c: i for i, c in enumerate(CLASSES)}
    cls_to_idx_fp.update({"FP_" + c: i + len(CLASSES) for i, c in enumerate(CLASSES)})
    cfg = dict(CLASSES=CLASSES, cls_to_idx=cls_to_idx, cls_to_idx_fp=cls_to_idx_fp, tracking_classes=TRACKING,
               tracking_classes_fp=TRACKING, subsample_sparse=subsample_sparse, subsample_dense=subsample_dense,
               return_mode="dict", validation_seed=seed,
               sparse_loader=dict(type="ToySparseLoader", metadata=dict(obj_infos=obj_infos), data_root=root,
                                  load_feats=["xyz"], load_dims=[3], load_fraction=1.0, tracking_classes=TRACKING,
                                  min_points=1, use_distance=(kind == "val_even"), filter_mode="pts"),
               complete_loader=dict(type="FakeCompleteLoader", subsample_num=subsample_dense))
    rn = R.reidentification_nuscenes
    cwd = os.getcwd()
    os.chdir(scratch)
    try:
        np.random.seed(seed)
        if kind == "train":
            return rn.ReIDDatasetNuscenesFP(**cfg)
        if kind == "val":
            return rn.ReIDDatasetNuscenesFPVal(max_combinations, **cfg)
        return rn.ReIDDatasetNuscenesFPValEven(max_combinations, True, **cfg)
    finally:
        os.chdir(cwd)
