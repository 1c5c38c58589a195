"""`ReIDNet`: siamese point-cloud re-identification model with the reference's model-level plugin
API (bentherien/point-cloud-reid, mmdet3d/models/ReIDNet.py:40-96 module_obj/build_module,
:111-776 ReIDNet), so that configs_reid/* build unchanged through FUSIONMODELS / build_model.

What runs where: both clouds of every pair go through the backbone in one batch
(siamese_forward, ref :311-332); the two-stage bidirectional cross-attention (xcorr_eff, ref
:231-247) runs on all 2B clouds at once with a partner index instead of four separate calls;
'point-cat' + pool 'both' + match head (ref :526-534, :444-462) is a single launch.  All of it
is libpcr_hip.so; this file only owns parameters, shapes and the mmdet-style entry points.
"""
import copy
from collections import OrderedDict

import torch
import torch.nn as nn

from pcr_amd import engine
from pcr_amd import _lib as L
from pcr_amd.lazylog import LazyScalars
from .attention import corss_attention, cross_lin_attn, local_self_attention
from .backbone_net import Pointnet_Backbone
from .dgcnn_orig import DGCNN
from .builder import FUSIONMODELS
from .lanegcn_nets import LinearRes
from .pointnet import PointNet
from .pointnet2_ssg import PointNet2SSG


class _OutOfScope(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("%s is outside the siamese point-cloud hot path rebuilt here "
                                  "(SURVEY.md section 8f)" % type(self).__name__)


class PostRes(_OutOfScope):
    pass


# type names a config may use (reference ReIDNet.py:40-75)
module_obj = {
    "Linear": nn.Linear,
    "ReLU": nn.ReLU,
    "LSTM": nn.LSTM,
    "GroupNorm": nn.GroupNorm,
    "Embedding": nn.Embedding,
    "LayerNorm": nn.LayerNorm,
    "PostRes": PostRes,
    "LinearRes": LinearRes,
    "Pointnet_Backbone": Pointnet_Backbone,
    "corss_attention": corss_attention,
    "local_self_attention": local_self_attention,
    "Conv1d": nn.Conv1d,
    "Conv2d": nn.Conv2d,
    "BatchNorm1d": nn.BatchNorm1d,
    "Sigmoid": nn.Sigmoid,
    "cross_lin_attn": cross_lin_attn,
    "dgcnn": DGCNN,
    "PointNet": PointNet,
    # not in the reference table: BASELINE config 2's build-defined PointNet++ SSG encoder
    "PointNet2SSG": PointNet2SSG,
}


def build_module(cfg):
    """None / {} -> None, list -> nn.Sequential, dict -> module_obj[type](**rest).
    Like the reference (:77-96) this consumes the 'type' key of the dict it is given."""
    if cfg is None or cfg == {}:
        return None
    if isinstance(cfg, list):
        return build_sequential(cfg)
    cls_ = module_obj[cfg["type"]]
    del cfg["type"]
    return cls_(**cfg)


def build_sequential(module_list):
    if module_list is None or module_list == {}:
        return None
    return nn.Sequential(*[build_module(cfg) for cfg in module_list])


def get_accuracy(y_true, y_prob):
    assert y_true.ndim == 1 and y_true.size() == y_prob.size()
    y_prob = y_prob > 0.5
    return (y_true == y_prob).sum().item() / y_true.size(0)


@FUSIONMODELS.register_module()
class ReIDNet(nn.Module):
    def __init__(self, hidden_size, backbone, cls_head, match_head, shape_head, fp_head, downsample,
                 cross_stage1, local_stage1, cross_stage2, local_stage2, match_type="xcorr", pool_type="max",
                 combine="cat", compute_summary=True, train_cfg=None, test_cfg=None,
                 backbone_list=[512, 256, 128], use_dgcnn=False,
                 losses_to_use=dict(kl=True, match=True, cls=True, shape=True, fp=True, dense=False),
                 output_sequence_size=32, alpha=dict(kl=1, match=1, cls=1, shape=1, fp=1, triplet=1, dense=1),
                 triplet_sample_num=5, triplet_loss=dict(margin=0.2, p=2), eval_only=False, use_o=False,
                 eval_flip=False):
        super().__init__()
        self.eval_only = eval_only
        self.hidden_size = hidden_size
        self.match_type = match_type
        self.backbone = build_module(backbone)
        self.cls_head = build_module(cls_head)
        self.match_head = build_module(match_head)
        self.shape_head = build_module(shape_head)
        self.fp_head = build_module(fp_head)
        self.downsample = build_module(downsample)
        self.cross_stage1 = build_module(cross_stage1)
        self.local_stage1 = build_module(local_stage1)
        self.cross_stage2 = build_module(cross_stage2)
        self.local_stage2 = build_module(local_stage2)

        self.losses_to_use = dict(kl=False, match=True, cls=False, shape=False, fp=False, dense=False)
        self.losses_to_use.update(losses_to_use)
        self.backbone_list = backbone_list
        self.output_sequence_size = output_sequence_size
        self.pool_type = pool_type
        self.bce = nn.BCEWithLogitsLoss()
        self.alpha = alpha
        self.use_o = use_o
        self.eval_flip = eval_flip
        self.verbose = False
        self.sampling = None
        self.compute_summary = compute_summary
        self.use_dgcnn = use_dgcnn
        self.combine = combine
        self.triplet_sample_num = triplet_sample_num
        self._head_plan = None
        self._head_key = None

    # ------------------------------------------------------------------ inputs
    @staticmethod
    def _stack(*lists):
        return [torch.stack(x, dim=0) for x in lists]

    @staticmethod
    def _cat(*lists):
        return [torch.cat(x, dim=0) for x in lists]

    def preprocess_inputs(self, sparse_1, sparse_2, dense_1, dense_2, label_1, label_2, id_1, id_2):
        sparse_1, sparse_2, dense_1, dense_2 = self._stack(sparse_1, sparse_2, dense_1, dense_2)
        label_1, label_2, id_1, id_2 = self._cat(label_1, label_2, id_1, id_2)
        if self.eval_flip:
            return sparse_2, sparse_1, dense_2, dense_1, label_2, label_1, id_2, id_1
        return sparse_1, sparse_2, dense_1, dense_2, label_1, label_2, id_1, id_2

    def preprocess_inputs_size(self, sparse_1, sparse_2, dense_1, dense_2, label_1, label_2, id_1, id_2,
                               size_1, size_2):
        sparse_1, sparse_2, dense_1, dense_2 = self._stack(sparse_1, sparse_2, dense_1, dense_2)
        return (sparse_1, sparse_2, dense_1, dense_2,
                *self._cat(label_1, label_2, id_1, id_2, size_1, size_2))

    def preprocess_inputs_size_vis(self, sparse_1, sparse_2, dense_1, dense_2, label_1, label_2, id_1, id_2,
                                   size_1, size_2, vis_1, vis_2):
        sparse_1, sparse_2, dense_1, dense_2 = self._stack(sparse_1, sparse_2, dense_1, dense_2)
        return (sparse_1, sparse_2, dense_1, dense_2,
                *self._cat(label_1, label_2, id_1, id_2, size_1, size_2, vis_1, vis_2))

    # ------------------------------------------------------------------ split-bf16 guard (pcr_amd/engine.py)
    def _weights_key(self, full=True):
        """identity + version of every parameter and buffer.  full: one walk over the module tree (~150 us for the ~230
        tensors of a ReID model; every guard tick and guard_state() do it), so that REPLACED Parameters
        (load_state_dict(assign=True), m.weight = nn.Parameter(..), re-registered buffers) are seen, not only in-place
        writes; not full: the versions of the tensors found by the last walk (~25 us; precision_level(), several times
        per pass between two ticks)"""
        ts = None if full else self.__dict__.get("_pcr_tensors")
        if ts is None:
            tensors, h, stack = [], 0, [self]
            while stack:
                mod = stack.pop()
                for d in (mod._parameters, mod._buffers):
                    for t in d.values():
                        if t is not None:
                            tensors.append(t)
                            h = ((h * 1000003) ^ id(t)) & 0xFFFFFFFFFFFFFFFF
                for c in mod._modules.values():
                    if c is not None:
                        stack.append(c)
            ts = (tensors, h)
            self.__dict__["_pcr_tensors"] = ts
        v = 0
        for t in ts[0]:
            v += t._version
        return (v, ts[1], len(ts[0]))

    def _guard_active(self):
        return engine.GUARD and engine.PRECISION == "bf16x3" and not self.training

    def precision_level(self):
        """guard level of these weights: the calibrated one (0 before calibration -- every inference entry point
        calibrates on its first batch, _guard_tick / _match_level, so an uncalibrated 0 is only seen inside a HIP-graph
        capture or on host tensors; always 0 in training mode, with the guard off, or outside "bf16x3" mode)"""
        if not self._guard_active():
            return 0
        st = self.__dict__.get("_pcr_guard")
        return st["level"] if st is not None and st["key"] == self._weights_key(full=False) else 0

    def guard_state(self):
        """what calibrate_precision (and the sentinel since) measured for the current weights, or None"""
        st = self.__dict__.get("_pcr_guard")
        if st is None or st["key"] != self._weights_key():
            return None
        return {k: v for k, v in st.items() if k != "key"}

    def _hot(self, s1, s2):
        xyz1, xyz2, h1, h2 = self._siamese_forward(s1, s2)
        return self._match_logits(h1, h2, xyz1, xyz2, inference=True)[0]

    def calibrate_precision(self, sparse_1, sparse_2, bound=None, max_pairs=64):
        """split-bf16 guard: run the hot path on (up to max_pairs of) this batch in f32 and at guard levels 0, 1, 2, keep the
        first level whose logits stay within engine.GUARD_ACCEPT x `bound` (engine.GUARD_BOUND, half the 1e-4 parity bound) of the f32 path's,
        for as long as the weights do not change (the sentinel re-checks it on live batches: _guard_tick).
        -> {"level", "dlogit": {level: max |logit - f32 logit|}, ...}"""
        bound = engine.GUARD_BOUND if bound is None else float(bound)
        s1, s2 = sparse_1[:max_pairs].contiguous(), sparse_2[:max_pairs].contiguous()
        self.__dict__.pop("_pcr_guard", None)
        dl, level = {}, 0
        if engine.PRECISION == "bf16x3" and not self.training:
            self.__dict__["_pcr_guard_busy"] = True
            prof, engine.PROFILE = engine.PROFILE, None          # (the guard's own passes are not part of a profiled pass)
            try:
                with torch.no_grad():
                    with engine.precision("f32"):
                        ref = self._hot(s1, s2)
                    for level in (0, 1, 2):
                        with engine.guard_level(level):
                            dl[level] = float((self._hot(s1, s2) - ref).abs().max())
                        if dl[level] <= bound * engine.GUARD_ACCEPT:      # (a margin for the next batch: engine.GUARD_ACCEPT)
                            break
            finally:
                self.__dict__["_pcr_guard_busy"] = False
                engine.PROFILE = prof
        st = dict(key=self._weights_key(), level=level, dlogit=dl, bound=bound, pairs=int(s1.shape[0]), calls=0,
                  sentinel=dict(every=engine.GUARD_EVERY, checks=0, worst=0.0, raised=[]))
        self.__dict__["_pcr_guard"] = st
        return self.guard_state()

    def _sentinel(self, c1, c2, st):
        """run-time re-check of the calibrated level on (up to engine.GUARD_SENTINEL_PAIRS of) the LIVE batch: the f32 path
        against the current level; on a breach of the bound the level of this weight version is raised to the first
        one that holds (2 = the f32 path always does) BEFORE the batch itself is computed, and the event is logged.
        The sampled pairs rotate through the batch from check to check."""
        sen = st["sentinel"]
        m = min(engine.GUARD_SENTINEL_PAIRS, int(c1.shape[0]))
        lo = (sen["checks"] * m) % max(1, int(c1.shape[0]) - m + 1)
        s1, s2 = c1[lo:lo + m].contiguous(), c2[lo:lo + m].contiguous()
        self.__dict__["_pcr_guard_busy"] = True
        prof, engine.PROFILE = engine.PROFILE, None
        try:
            with torch.no_grad():
                with engine.precision("f32"):
                    ref = self._hot(s1, s2)
                level = st["level"]
                while True:
                    with engine.guard_level(level):
                        d = float((self._hot(s1, s2) - ref).abs().max()) if level < 2 else 0.0
                    if d <= st["bound"] or level >= 2:
                        break
                    level += 1
        finally:
            self.__dict__["_pcr_guard_busy"] = False
            engine.PROFILE = prof
        sen["checks"] += 1
        if level != st["level"]:
            import logging
            logging.getLogger("pcr_amd.guard").warning(
                "split-bf16 guard: live batch %d deviates from the f32 path beyond %.1e at level %d; these weights run "
                "at level %d from here on", st["calls"], st["bound"], st["level"], level)
            sen["raised"].append(dict(call=st["calls"], was=st["level"], now=level))
            st["level"] = level
        else:
            sen["worst"] = max(sen["worst"], d)

    def _guard_tick(self, c1, c2):
        """top of every inference entry point that sees raw clouds (c1, c2: (B,N,3) halves of pairs, or the two halves of a
        gallery): lazy calibration on the first batch of a weight version (the tracker entry points ran
        unguarded until someone called calibrate_precision by hand), then the sentinel every engine.GUARD_EVERY batches.
        Inside a HIP-graph capture nothing can be measured (host reads): the captured launches keep the level the model
        has -- calibrate before capturing, as bench.py does."""
        if not self._guard_active() or self.__dict__.get("_pcr_guard_busy") or not c1.is_cuda or c1.shape[0] == 0:
            return
        if torch.cuda.is_current_stream_capturing():
            return
        st = self.__dict__.get("_pcr_guard")
        if st is None or st["key"] != self._weights_key():
            self.calibrate_precision(c1, c2)
            return
        st["calls"] += 1
        every = engine.GUARD_EVERY
        if every > 0 and st["level"] < 2 and st["calls"] % every == 0:
            self._sentinel(c1, c2, st)

    @staticmethod
    def _halves(pts):
        """a batch of single clouds (forward_inference) as pairs for the guard: first half against second half"""
        m = pts.shape[0] // 2
        return (pts[:m], pts[m:2 * m]) if m else (pts, pts)

    def _raw_clouds(self, xyz1, xyz2):
        """match-only entry (features computed elsewhere): the raw clouds are what the point-major backbones return as
        xyz; PointNet / DGCNN hand back a transformed, channel-major xyz that cannot be re-encoded"""
        if self.use_dgcnn or isinstance(self.backbone, PointNet):
            return None
        return xyz1, xyz2

    # ------------------------------------------------------------------ encoder
    def forward_inference(self, pts_batched):
        # (PointNet / DGCNN take channel-major clouds (B,3,N); the guard's pairs are point-major like siamese_forward's)
        cm = self.use_dgcnn or isinstance(self.backbone, PointNet)
        self._guard_tick(*self._halves(pts_batched.permute(0, 2, 1) if cm else pts_batched))
        lvl = self.precision_level()
        with torch.no_grad():
            if lvl:
                with engine.guard_level(lvl):
                    return self.backbone(pts_batched, self.backbone_list)
            return self.backbone(pts_batched, self.backbone_list)

    def siamese_forward(self, sparse_1, sparse_2):
        self._guard_tick(sparse_1, sparse_2)
        lvl = self.precision_level()
        if lvl:
            with engine.guard_level(lvl):
                return self._siamese_forward(sparse_1, sparse_2)
        return self._siamese_forward(sparse_1, sparse_2)

    def _siamese_forward(self, sparse_1, sparse_2):
        """(B,N,3) x 2 -> xyz1, xyz2 (B,N,3), h1, h2 (B,C,N); one backbone pass over 2B clouds"""
        assert sparse_1.shape == sparse_2.shape
        b, num_points, _ = sparse_1.shape
        both = torch.cat([sparse_1, sparse_2], dim=0)
        if self.use_dgcnn or isinstance(self.backbone, PointNet):
            # per-point 1024-d encoder features, reduced per point by `downsample` (ref :316-324)
            xyz, h = self.backbone(both.permute(0, 2, 1).contiguous(), self.backbone_list)
            if self.downsample is not None:
                if self.training:
                    from pcr_amd import train_graph
                    h = train_graph.downsample_points(self.downsample, h)
                else:
                    from pcr_amd import rows
                    h = rows.downsample_points(self.downsample, h)
            xyz = xyz.permute(0, 2, 1)
            return xyz[:b], xyz[b:], h[:b], h[b:]
        xyz, h = self.backbone(both, self.backbone_list)
        return xyz[:b], xyz[b:], h[:b], h[b:]

    # ------------------------------------------------------------------ matching
    @staticmethod
    def _pair_batch(a1, a2):
        """two halves -> one (2B, ...) tensor, without a copy when they already are adjacent views"""
        if (a1.is_contiguous() and a2.is_contiguous() and a1.untyped_storage().data_ptr() == a2.untyped_storage().data_ptr()
                and a2.storage_offset() == a1.storage_offset() + a1.numel()):
            return a1.as_strided((2 * a1.shape[0],) + tuple(a1.shape[1:]), a1.stride(), a1.storage_offset())
        return torch.cat([a1, a2], dim=0)

    def _xcorr_eff_batched(self, h1, xyz1, h2, xyz2):
        b = h1.shape[0]
        feats = self._pair_batch(h1, h2).contiguous()
        xyz = self._pair_batch(xyz1, xyz2).contiguous()
        # (built on the device: a pageable host-to-device copy here stalls the stream -- measured on the gallery path,
        # where the same 256 KB index copy cost 45 ms every third step)
        partner = torch.cat([torch.arange(b, 2 * b, device=feats.device, dtype=torch.int32),
                             torch.arange(0, b, device=feats.device, dtype=torch.int32)])
        s1 = self.cross_stage1.forward_paired(feats, xyz, partner)
        return self.cross_stage2.forward_paired(s1, xyz, partner)     # (2B,C,N): [o1; o2]

    def xcorr_eff(self, o1, xyz1, o2, xyz2, combine="add"):
        b = o1.shape[0]
        o = self._xcorr_eff_batched(o1, xyz1, o2, xyz2)
        o1, o2 = o[:b], o[b:]
        if self.combine == "add":
            out = o1 + o2
        elif self.combine == "minus":
            out = o1 - o2
        elif self.combine == "cat":
            out = torch.cat([o1, o2], dim=1)
        elif self.combine == "point-cat":
            out = torch.cat([o1, o2], dim=2)
        else:
            raise NotImplementedError(self.combine)
        return out, o1, o2

    def _head(self, device):
        if not (isinstance(self.match_head, nn.Sequential) and len(self.match_head) == 2
                and isinstance(self.match_head[0], LinearRes) and isinstance(self.match_head[1], nn.Linear)):
            raise L.PcrError("fused match head expects [LinearRes, Linear] as in every ReID config")
        key = (str(device), engine.param_version(self.match_head))
        if self._head_key != key:
            self._head_plan = engine.HeadPlan(self.match_head[0], self.match_head[1], device)
            self._head_key = key
        return self._head_plan

    def get_pooled_feats(self, h_cat):
        from pcr_amd import rows
        if self.pool_type == "max":
            # nn.MaxPool1d(output_sequence_size) on the permuted (B,N,C) tensor: a max over channel windows (reference
            # :145, :526-528) -- (B,N) when C == output_sequence_size, as in reid_pts_point-transformer_baseline.py
            return rows.pool_channel_max(h_cat, self.output_sequence_size)
        if self.pool_type == "both":
            return rows.pool_both(h_cat)
        raise NotImplementedError("pool_type=%r" % self.pool_type)

    def _fused_matching(self):
        return self.match_type == "xcorr_eff" and self.combine == "point-cat" and self.pool_type == "both"

    def _head_rows(self, pooled):
        """match_head ([LinearRes..., Linear]) on pooled rows (B,F) through the row kernels (pcr_amd/rows.py)"""
        from pcr_amd import rows
        x = pooled.t().contiguous().unsqueeze(0)                 # (1,F,B): channel-major, samples as tokens
        return rows.downsample_points(self.match_head, x).reshape(-1)

    def xcorr(self, search_feat, search_xyz, template_feat, template_xyz):
        """cross -> local -> cross -> local on the search branch (reference ReIDNet.py:250-256)"""
        a = self.cross_stage1(search_feat, search_xyz, template_feat, template_xyz)
        b = self.local_stage1(a, search_xyz)
        c = self.cross_stage2(b, search_xyz, template_feat, template_xyz)
        return self.local_stage2(c, search_xyz)

    def xcorr_baseline(self, search_feat, search_xyz, template_feat, template_xyz):
        a = self.cross_stage1(search_feat, search_xyz, template_feat, template_xyz)
        return self.cross_stage2(a, search_xyz, template_feat, template_xyz)

    def _match_logits(self, h1, h2, xyz1, xyz2, inference=False):
        """logits (B) and the stage-2 features; fused single-launch tail for the configuration every point
        ReID config uses, generic composition (reference ReIDNet.py:387-462) for the other variants"""
        if self.training:
            # training mode: the differentiable HIP graph (raises PcrError for the matchings it does not cover -- the
            # inference compositions below have no backward)
            from pcr_amd import train_graph
            return train_graph.match_logits(self, h1, xyz1, h2, xyz2)
        if self._fused_matching():
            o = self._xcorr_eff_batched(h1, xyz1, h2, xyz2)
            return self._head(o.device).run(o), o
        if self.match_type == "xcorr_eff":
            match_in, o1, o2 = self.xcorr_eff(h1, xyz1, h2, xyz2, self.combine)
            return self._head_rows(self.get_pooled_feats(match_in)), torch.cat([o1, o2], dim=0)
        if self.match_type == "xcorr":
            if self.local_stage1 is None or self.local_stage2 is None:
                raise L.PcrError("match_type='xcorr' needs local_stage1 / local_stage2 (local_self_attention)")
            match_in = self.xcorr(h1, xyz1, h2, xyz2)
            return self._head_rows(self.get_pooled_feats(match_in)), None
        if self.match_type == "xcorr-baseline":
            match_in = self.xcorr_baseline(h1, xyz1, h2, xyz2)
            return self._head_rows(self.get_pooled_feats(match_in)), None
        if self.match_type == "concat":
            if inference:     # the reference's inference entry point pools with self.maxpool whatever pool_type (:455-457)
                from pcr_amd import rows
                pool = lambda h: rows.pool_channel_max(h, self.output_sequence_size)
            else:
                pool = self.get_pooled_feats
            cat = torch.cat([pool(h1), pool(h2)], dim=1)
            return self._head_rows(cat), None
        raise NotImplementedError("match_type=%r" % self.match_type)

    def _match_level(self, xyz1, xyz2):
        """level for an entry point that only sees features: the calibrated one; an uncalibrated weight version is
        calibrated from the raw clouds (xyz of the point-major backbones) or, where they cannot be recovered, runs the
        matching in f32 (level 2) -- never unguarded"""
        if not self._guard_active() or self.__dict__.get("_pcr_guard_busy"):
            return self.precision_level()
        st = self.__dict__.get("_pcr_guard")
        if st is not None and st["key"] == self._weights_key(full=False):
            return st["level"]
        if self.guard_state() is None and xyz1.is_cuda and not torch.cuda.is_current_stream_capturing():
            raw = self._raw_clouds(xyz1, xyz2)
            if raw is None or raw[0].shape[0] == 0:
                return 2
            self.calibrate_precision(*raw)
        return self.precision_level()

    def match_forward_inference(self, h1, h2, xyz1, xyz2):
        lvl = self._match_level(xyz1, xyz2)
        if lvl:
            with engine.guard_level(lvl):
                return self._match_logits(h1, h2, xyz1, xyz2, inference=True)[0]
        return self._match_logits(h1, h2, xyz1, xyz2, inference=True)[0]

    def match_gallery(self, h, xyz, pairs):
        lvl = self._match_level(*self._halves(xyz))
        if lvl:
            with engine.guard_level(lvl):
                return self._match_gallery(h, xyz, pairs)
        return self._match_gallery(h, xyz, pairs)

    def _match_gallery(self, h, xyz, pairs):
        """Amortised matching (the reference's tracker use-case of forward_inference +
        match_forward_inference, ReIDNet.py:189-191, 444-462): every object is encoded ONCE, then any list
        of (i, j) combinations is scored.  h (M,C,N), xyz (M,N,3) from forward_inference / siamese_forward;
        pairs (P,2) integer tensor -> logits (P), identical to match_forward_inference(h[i], h[j], ...).
        Stage-1 key/value state is computed once per object and shared by all its pairs."""
        if self.match_type != "xcorr_eff" or self.combine != "point-cat" or self.pool_type != "both":
            raise NotImplementedError("match_gallery covers the xcorr_eff / point-cat / both matching head")
        h, xyz = h.contiguous(), xyz.contiguous()
        n_pts = h.shape[2]
        p1 = self.cross_stage1.plan(h.device)
        p2 = self.cross_stage2.plan(h.device)
        kv1 = p1.kv(h, xyz)                                  # stage-1 key/value state: once per OBJECT
        # round 6: the stage-2 apply launch leaves every virtual cloud's per-channel maximum and sum (512 bytes) instead of
        # its 32 KB output for pool_head to read back; the head then runs over the pooled ROWS on the matrix core
        # (pcr_attn_apply_pool_ok: the launch shape and the arithmetic decide, never the number of pairs)
        pooled_ok = p2.pool_ok(n_pts, n_pts)
        out = []
        # the launches index clouds by a 16-bit grid dimension: at most 32 k pairs (64 k virtual clouds) per pass
        chunk = 32000
        for lo in range(0, pairs.shape[0], chunk):
            pc = pairs[lo:lo + chunk]
            n_pairs = pc.shape[0]
            i = pc[:, 0].to(device=h.device, dtype=torch.int32)
            j = pc[:, 1].to(device=h.device, dtype=torch.int32)
            q_idx = torch.cat([i, j]).contiguous()        # virtual cloud b < P: object i queries object j ...
            k_idx = torch.cat([j, i]).contiguous()        # ... and b >= P: object j queries object i
            s1 = p1.apply(h, None, kv1, n_pts, kv_index=k_idx, q_index=q_idx, n_out=2 * n_pairs)
            xyz_v = xyz.index_select(0, q_idx.long()).contiguous()
            partner = torch.cat([torch.arange(n_pairs, 2 * n_pairs, device=h.device, dtype=torch.int32),
                                 torch.arange(0, n_pairs, device=h.device, dtype=torch.int32)])
            if pooled_ok:
                pl = p2.apply(s1, None, p2.kv(s1, xyz_v), n_pts, kv_index=partner, pooled=True)     # (2n, 2, C): [max | sum]
                a, b2 = pl[:n_pairs], pl[n_pairs:]
                feat = torch.cat([torch.maximum(a[:, 0], b2[:, 0]), (a[:, 1] + b2[:, 1]) / float(2 * n_pts)], dim=1)
                out.append(self._head_rows(feat))
            else:
                o = p2.apply(s1, None, p2.kv(s1, xyz_v), n_pts, kv_index=partner)
                out.append(self._head(o.device).run(o))
        if not out:
            return torch.empty(0, dtype=torch.float32, device=h.device)
        return out[0] if len(out) == 1 else torch.cat(out, dim=0)

    def get_match_supervision(self, h1, h2, xyz1, xyz2, id_1, id_2):
        return h1, h2, xyz1, xyz2, (id_1 == id_2).float()

    def match_forward(self, h1, h2, xyz1, xyz2, match, log_vars, device, prefix=""):
        if not self.losses_to_use["match"]:
            return None, torch.tensor(0.0, requires_grad=True, device=device), (None, None)
        b = h1.shape[0]
        lvl = self.precision_level()
        if lvl:
            with engine.guard_level(lvl):
                match_preds, o = self._match_logits(h1, h2, xyz1, xyz2)
        else:
            match_preds, o = self._match_logits(h1, h2, xyz1, xyz2)
        match_loss = self.bce(match_preds, match) * self.alpha["match"]
        if o is None:
            o = [None] * (2 * b)
        if self.compute_summary and log_vars is not None:
            # the reference's six log entries (ReIDNet.py:426-435: `.item()` each), computed on the device as ONE small
            # tensor and read lazily (pcr_amd/lazylog.py): same keys -- including the reference's swapped
            # num_preds / num_gt naming -- and same values, no host round trip inside the step
            pred = match_preds.detach() > 0                      # sigmoid(x) > 0.5
            m = match.detach()
            n = float(m.numel())
            gt1, p1 = m.sum(), pred.float().sum()
            stats = torch.stack([match_loss.detach().reshape(()), pred.float().eq(m).float().mean(),
                                 n - gt1, gt1, n - p1, p1])
            names = [prefix + k for k in ("match_loss", "match_acc", "num_preds_0", "num_preds_1", "num_gt_0", "num_gt_1")]
            ints = [False, False, True, True, True, True]
            if isinstance(log_vars, LazyScalars):
                log_vars.add_device(names, stats, ints)
            else:
                for k, i, v in zip(names, ints, stats.tolist()):
                    log_vars[k] = int(round(v)) if i else v
        return match_preds, match_loss, (o[:b], o[b:])

    def _check_losses(self):
        # the constructor default enables every auxiliary loss (as in the reference), but every ReID
        # config switches them off and builds no heads for them; only the match loss is on the hot path
        for k in ("kl", "cls", "shape", "fp", "dense", "triplet"):
            if self.losses_to_use.get(k, False):
                raise NotImplementedError("loss '%s' is not part of the siamese matching hot path; the ReID "
                                          "configs train/evaluate with match only (SURVEY.md section 8)" % k)

    # ------------------------------------------------------------------ mmdet-style entry points
    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(**kwargs)
        return self.forward_test(**kwargs)

    def forward_train(self, sparse_1, sparse_2, dense_1, dense_2, label_1, label_2, id_1, id_2):
        if self.eval_only:
            exit(0)     # reference behaviour: testing configs stop at the first training iteration (:587-588)
        self._check_losses()
        log_vars, losses = LazyScalars(), {}
        sparse_1, sparse_2, dense_1, dense_2, label_1, label_2, id_1, id_2 = self.preprocess_inputs(
            sparse_1, sparse_2, dense_1, dense_2, label_1, label_2, id_1, id_2)
        device = sparse_1.device
        xyz1, xyz2, h1, h2 = self.siamese_forward(sparse_1, sparse_2)
        h1, h2, xyz1, xyz2, match = self.get_match_supervision(h1, h2, xyz1, xyz2, id_1, id_2)
        _, match_loss, _ = self.match_forward(h1, h2, xyz1, xyz2, match, log_vars, device)
        losses["reid_loss"] = match_loss
        return losses, log_vars

    def forward_test(self, sparse_1, sparse_2, dense_1, dense_2, label_1, label_2, id_1, id_2, size_1, size_2,
                     vis_1, vis_2, *args, **kwargs):
        self._check_losses()
        (sparse_1, sparse_2, dense_1, dense_2, label_1, label_2, id_1, id_2, size_1, size_2, vis_1,
         vis_2) = self.preprocess_inputs_size_vis(sparse_1, sparse_2, dense_1, dense_2, label_1, label_2, id_1,
                                                  id_2, size_1, size_2, vis_1, vis_2)
        device = sparse_1.device
        # (siamese_forward calibrates the split-bf16 guard on the first batch after the weights changed and re-checks it
        # every engine.GUARD_EVERY batches: _guard_tick)
        xyz1, xyz2, h1, h2 = self.siamese_forward(sparse_1, sparse_2)
        h1, h2, xyz1, xyz2, match = self.get_match_supervision(h1, h2, xyz1, xyz2, id_1, id_2)
        match_preds, match_loss, _ = self.match_forward(h1, h2, xyz1, xyz2, match, None, device)
        labels = torch.cat([label_1, label_2], dim=0)
        zero = torch.tensor([0.0])
        results = OrderedDict()
        results["val_dense_loss"] = zero.clone()
        results["val_fp_loss"] = zero.clone()
        results["val_match_loss"] = torch.tensor([match_loss])
        results["val_shape_loss"] = zero.clone()
        results["val_cls_loss"] = zero.clone()
        results["val_kl_loss"] = zero.clone()
        results["val_match_preds"] = match_preds
        results["val_match_gt"] = match
        results["val_cls_preds"] = None
        results["val_cls_gt"] = labels
        results["val_fp_preds"] = None
        results["val_fp_gt"] = (labels > 9).float()
        results["match_classes"] = torch.cat([label_1.unsqueeze(1), label_2.unsqueeze(1)], dim=1)
        results["is_fp"] = torch.logical_or(label_1 > 9, label_2 > 9)
        results["num_points"] = torch.cat([size_1.unsqueeze(1), size_2.unsqueeze(1)], dim=1)
        results["val_vis_gt_all"] = torch.cat([vis_1.unsqueeze(1), vis_2.unsqueeze(1)], dim=1)
        return [results]

    @staticmethod
    def _parse_losses(losses):
        """mmdet BaseDetector._parse_losses: mean every entry, sum those whose key contains 'loss',
        all-reduce the logged scalars across ranks when torch.distributed is initialised."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError("%s is not a tensor or list of tensors" % name)
        loss = sum(v for k, v in log_vars.items() if "loss" in k)
        log_vars["loss"] = loss
        # mmdet reduces and reads every entry on its own (one all-reduce + one .item() each); here the entries travel as
        # ONE stacked tensor: one all-reduce, one asynchronous copy, read on first access (pcr_amd/lazylog.py)
        import torch.distributed as dist
        from pcr_amd import lazylog
        names = list(log_vars.keys())
        stacked = torch.stack([log_vars[k].detach().reshape(()).float() for k in names])
        log_vars = LazyScalars()
        if dist.is_available() and dist.is_initialized():
            if lazylog.DEFER_REDUCE:
                # under pcr_amd.train.Trainer: averaged in the tail of the gradient bucket -- no collective inside the
                # (capturable) forward + backward, one all-reduce per iteration in total
                log_vars.add_device(names, stacked, reduce=True)
                return loss, log_vars
            stacked = stacked.clone()
            dist.all_reduce(stacked.div_(dist.get_world_size()))
        log_vars.add_device(names, stacked)
        return loss, log_vars

    def train_step(self, data, optimizer):
        losses, log_vars_train = self(**data)
        loss, log_vars = self._parse_losses(losses)
        log_vars.update(log_vars_train)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data["sparse_1"]))

    def val_step(self, data, optimizer=None):
        losses, _ = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data["sparse_1"]))

    def extract_feat(self, *args, **kwargs):
        raise NotImplementedError

    def show_result(self):
        raise NotImplementedError

    def aug_test(self, *args, **kwargs):
        raise NotImplementedError

    def simple_test(self, *args, **kwargs):
        raise NotImplementedError

    def init_weights(self):
        pass
