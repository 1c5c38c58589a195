"""Host side of the training-mode kernels (include/pcr.h section C; csrc/train_kernels.hip, train_sa_kernels.hip):
ctypes parameter blocks, launches, and the torch.autograd.Functions that put them under `loss.backward()`.

What runs where: every dense layer of the training graph -- the grouped set-abstraction MLPs with BatchNorm in
batch-statistics mode (reference pointnet2_utils.py:333-360), the per-point tables of their first layer, the attention
projections / feed-forward layers and the match head -- is a HIP launch forward and a HIP launch backward on the
matrix core; autograd only strings the Functions together.  Nothing here has a CPU path.
"""
import ctypes
import os
import weakref

import torch
from torch.autograd import Function

from . import _lib as L
from .engine import _prof

c_fp = ctypes.c_void_p
FUSE_POOL = True       # SaEdgeTrain: take the max over K inside the last layer's launch where the library offers it


def _c32(n):
    return (n + 31) // 32 * 32


def _c8(n):
    return (n + 7) // 8 * 8


class _TFwd(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("cin1", ctypes.c_int), ("cin2", ctypes.c_int), ("cout", ctypes.c_int),
                ("L", ctypes.c_int), ("x", c_fp), ("x2", c_fp), ("isc", c_fp), ("ish", c_fp), ("in_relu", ctypes.c_int),
                ("wp", c_fp), ("bias", c_fp), ("res", c_fp), ("out_relu", ctypes.c_int), ("y", c_fp), ("stats", c_fp),
                ("pool_K", ctypes.c_int), ("pool_gamma", c_fp), ("pool_ymax", c_fp), ("pool_arg", c_fp)]


class _TBwd(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("cin1", ctypes.c_int), ("cin2", ctypes.c_int), ("cout", ctypes.c_int),
                ("L", ctypes.c_int), ("g", c_fp), ("y", c_fp), ("dy_mode", ctypes.c_int),
                ("ka", c_fp), ("kb", c_fp), ("kc", c_fp), ("argmax", c_fp), ("pooled", c_fp),
                ("K", ctypes.c_int), ("S", ctypes.c_int), ("x", c_fp), ("x2", c_fp),
                ("isc", c_fp), ("ish", c_fp), ("iinv", c_fp), ("in_relu", ctypes.c_int), ("wpT", c_fp),
                ("dx", c_fp), ("dx2", c_fp), ("dstats", c_fp), ("dwp", c_fp), ("dbp", c_fp), ("part_stride", ctypes.c_long),
                ("precision", ctypes.c_int), ("wpT_bf", c_fp)]


class _BnFwd(ctypes.Structure):
    _fields_ = [("part", c_fp), ("nparts", ctypes.c_int), ("C", ctypes.c_int), ("R", ctypes.c_double),
                ("gamma", c_fp), ("beta", c_fp), ("eps", ctypes.c_float), ("momentum", ctypes.c_float),
                ("running_mean", c_fp), ("running_var", c_fp),
                ("scale", c_fp), ("shift", c_fp), ("inv_scale", c_fp), ("mean", c_fp), ("invstd", c_fp),
                ("shift0", c_fp), ("shift0_stride", ctypes.c_int)]


class _BnBwd(ctypes.Structure):
    _fields_ = [("part", c_fp), ("nparts", ctypes.c_int), ("C", ctypes.c_int), ("R", ctypes.c_double),
                ("gamma", c_fp), ("mean", c_fp), ("invstd", c_fp),
                ("ka", c_fp), ("kb", c_fp), ("kc", c_fp), ("dgamma", c_fp), ("dbeta", c_fp), ("centre", c_fp)]


def _p(t):
    return None if t is None else t.data_ptr()


def _f32(*shape, device):
    return torch.empty(shape, dtype=torch.float32, device=device)


def _dev(t):
    L.require_cuda(t)
    L.require_f32(t)
    return t.contiguous()


# ------------------------------------------------------------------------------------------- launches --
class _Prepack:
    """Packed images (W and W^T) of every conv / linear weight of a model, refreshed by ONE launch per optimizer step
    (`prepack(model)`, called by the Trainer) instead of one small launch per layer.  Looked up by storage address +
    shape + Tensor._version: a weight that was not registered, or that changed since the refresh, simply misses and is
    packed on the spot."""

    def __init__(self):
        self.key = None
        self.entries = {}          # data_ptr -> [param, rows, cols, version, wp, wpT, wp_bf, wpT_bf]
        self.descs = None
        self.ndesc = 0
        self.params = []
        self.biases = {}
        self.want_bf = set()       # data_ptrs whose bf16 hi / lo images are refreshed too (asked for once: pack_both_bf)

    def build(self, params, biases):
        import numpy as np
        dev = params[0].device
        self.params = params
        old = self.entries
        self.entries = {}
        self.biases = {}           # id(bias) -> [param, version, zero-padded image]
        self.want_bf &= {p.data_ptr() for p in params}
        desc = np.zeros(len(params) + len(biases) + len(self.want_bf),
                        dtype=np.dtype([("w", "<u8"), ("out", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("kind", "<i4"),
                                        ("reserved", "<i4")]))
        nbf = 0
        for i, p in enumerate(params):
            rows, cols = p.shape[0], p.numel() // p.shape[0]
            n0, n1 = _c8(cols) * _c32(rows), _c8(rows) * _c32(cols)
            prev = old.get(p.data_ptr())
            if prev is not None and prev[0] is p and prev[1] == rows and prev[2] == cols:
                e = list(prev)         # (a rebuild that only adds bf images keeps the f32 images where they are)
                e[3] = -1
                out = e[4]._base if e[4]._base is not None else e[4]
            else:
                out = _f32(n0 + n1, device=dev)
                e = [p, rows, cols, -1, out[:n0], out[n0:], None, None]
            desc[i] = (p.data_ptr(), out.data_ptr(), rows, cols, 0, 0)
            if p.data_ptr() in self.want_bf:
                b0 = L.load().pcr_packed_weight_bf16_floats(rows, cols)
                b1 = L.load().pcr_packed_weight_bf16_floats(cols, rows)
                if e[6] is None:
                    ob = _f32(b0 + b1, device=dev)
                    e[6], e[7] = ob[:b0], ob[b0:]
                base = e[6]._base if e[6]._base is not None else e[6]
                desc[len(params) + len(biases) + nbf] = (p.data_ptr(), base.data_ptr(), rows, cols, 1, 0)
                nbf += 1
            else:
                e[6] = e[7] = None
            self.entries[p.data_ptr()] = e
        for i, b in enumerate(biases):          # cols = 0: a bias, copied into its zero-padded image by the same launch
            out = torch.zeros(_c32(b.numel()), dtype=torch.float32, device=dev)
            desc[len(params) + i] = (b.data_ptr(), out.data_ptr(), b.numel(), 0, 0, 0)
            self.biases[id(b)] = [b, -1, out, b.data_ptr()]
        self.ndesc = len(desc)
        self.descs = torch.from_numpy(desc.view(np.uint8).copy()).to(dev)
        self.key = tuple(id(p) for p in params) + tuple(id(b) for b in biases)

    def refresh(self, params, biases=(), build_only=False):
        params = [p for p in params if p.is_cuda and p.dim() >= 2 and p.dtype == torch.float32 and p.is_contiguous()]
        biases = [b for b in biases if b.is_cuda and b.dim() == 1 and b.dtype == torch.float32 and b.is_contiguous()]
        if not params:
            return
        if self.key != tuple(id(p) for p in params) + tuple(id(b) for b in biases) or \
                any(e[0].data_ptr() != k for k, e in self.entries.items()) or \
                any(e[0].data_ptr() != e[3] for e in self.biases.values()) or \
                any((k in self.want_bf) != (e[6] is not None) for k, e in self.entries.items()):
            self.build(params, biases)
            if build_only:
                return
        elif build_only or all(e[3] == e[0]._version for e in self.entries.values()) and \
                all(e[1] == e[0]._version for e in self.biases.values()):
            return        # nothing changed since the last refresh (micro-steps of a gradient accumulation)
        L.check(L.load().pcr_pack_weights_multi_f32(ctypes.c_void_p(self.descs.data_ptr()), self.ndesc,
                                                    L.stream_ptr()), "pcr_pack_weights_multi_f32")
        for e in self.entries.values():
            e[3] = e[0]._version
        for e in self.biases.values():
            e[1] = e[0]._version

    def lookup_bias(self, v, n):
        e = self.biases.get(id(v))
        if e is None or e[0] is not v or e[1] != v._version or e[2].numel() != _c32(n):
            return None
        return e[2]

    def lookup(self, w):
        e = self.entries.get(w.data_ptr())
        if e is None or e[3] != w._version or w.dim() != 2 or w.shape[0] != e[1] or w.shape[1] != e[2] or \
                not w.is_contiguous():
            return None
        return e[4], e[5]

    def lookup_bf(self, w):
        """-> (bf image of W, of W^T) | None (not registered / stale); registers the weight for the NEXT refresh"""
        e = self.entries.get(w.data_ptr())
        if e is None or w.dim() != 2 or w.shape[0] != e[1] or w.shape[1] != e[2] or not w.is_contiguous():
            return None
        self.want_bf.add(w.data_ptr())
        if e[6] is None or e[3] != w._version:
            return None
        return e[6], e[7]


# one table per model, owned by the model (dropped with it); `lookup` below searches the live tables
_PREPACKS = weakref.WeakKeyDictionary()


def _lookup(w):
    for t in _PREPACKS.values():
        hit = t.lookup(w)
        if hit is not None:
            return hit
    return None


def prepack(model, build_only=False):
    """refresh the packed images of every conv / linear weight of `model` in one launch (Trainer.step calls this once
    per iteration, after the previous update); pack_dev / pack_both then hit the cache.  Note for callers that keep an
    autograd graph across iterations: the images are overwritten in place by the next refresh.
    build_only: only (re)build the descriptor table and the image buffers if the set of weights or of requested bf16
    images changed -- allocations and a host-to-device copy, which a stream capture cannot take: the Trainer calls this
    right before it captures an iteration, whose own prepack() then is the one launch and nothing else."""
    mods = [m for m in model.modules()
            if isinstance(m, (torch.nn.Linear, torch.nn.Conv1d, torch.nn.Conv2d)) and m.weight is not None]
    tab = _PREPACKS.get(model)
    if tab is None:
        tab = _PREPACKS[model] = _Prepack()
    tab.refresh([m.weight for m in mods], [m.bias for m in mods if m.bias is not None], build_only=build_only)


def pack_dev(w, transpose=False):
    """(rows, cols) device matrix -> packed MFMA A-operand image of W or W^T (weights change every step, so the
    pack runs on the device; inference packs once on the host)"""
    hit = _lookup(w)
    if hit is not None:
        return hit[1] if transpose else hit[0]
    w = _dev(w.detach())
    rows, cols = w.shape
    cout, cin = (cols, rows) if transpose else (rows, cols)
    out = _f32(_c8(cin) * _c32(cout), device=w.device)
    L.check(L.load().pcr_pack_weight_dev_f32(L.ptr(w), rows, cols, cols, int(transpose), L.ptr(out), L.stream_ptr()),
            "pcr_pack_weight_dev_f32")
    return out


def pack_both(w):
    """-> (image of W, image of W^T) from ONE launch (forward and backward operands of a layer)"""
    hit = _lookup(w)
    if hit is not None:
        return hit
    w = _dev(w.detach())
    rows, cols = w.shape
    n0, n1 = _c8(cols) * _c32(rows), _c8(rows) * _c32(cols)
    out = _f32(n0 + n1, device=w.device)
    L.check(L.load().pcr_pack_weight_dev_f32(L.ptr(w), rows, cols, cols, 2, L.ptr(out), L.stream_ptr()),
            "pcr_pack_weight_dev_f32")
    return out[:n0], out[n0:]


_PAD_CACHE = {}


def pad32(v, n):
    """(n) vector -> zero-padded to a multiple of 32 (accumulator seeds are read 16 bytes at a time); cached per
    (tensor, version): a bias is padded once per optimizer step, not once per launch"""
    if v is None:
        return None
    for t in _PREPACKS.values():         # refreshed with the weights by the one launch per iteration
        hit = t.lookup_bias(v, n)
        if hit is not None:
            return hit
    # keyed by the tensor OBJECT (kept alive by the entry, so its id and storage cannot be recycled under the cache: a
    # (data_ptr, version) key returned another tensor's bias once the allocator had reused the address) and its version
    hit = _PAD_CACHE.get(id(v))
    if hit is not None and hit[0] is v and hit[2].numel() == _c32(n) and hit[2].device == v.device:
        if hit[1] != v._version:          # updated since (optimizer step): refresh the live part in place, one small copy
            hit[2][:n].copy_(v.detach())
            _PAD_CACHE[id(v)] = (v, v._version, hit[2])
        return hit[2]
    out = torch.zeros(_c32(n), dtype=torch.float32, device=v.device)
    out[:n] = v.detach()
    if len(_PAD_CACHE) > 256:
        _PAD_CACHE.clear()
    _PAD_CACHE[id(v)] = (v, v._version, out)
    return out


def groups(B, Ln):
    return L.load().pcr_train_groups(B, Ln)


def tdense_fwd(x, wp, cout, x2=None, isc=None, ish=None, in_relu=False, bias=None, res=None, out_relu=False,
               want_stats=False, pool=None):
    """-> (y, stats partials or None[, (ymax, argmax) or None when `pool` = (K, gamma) is given: the fused max over the K
    rows of every centre, where the launch can provide it (pcr_tdense_fwd_pooled)])"""
    x = _dev(x)
    B, cin1, Ln = x.shape
    cin2 = 0
    if x2 is not None:
        x2 = _dev(x2)
        cin2 = x2.shape[1]
    y = _f32(B, cout, Ln, device=x.device)
    p = _TFwd()
    p.B, p.cin1, p.cin2, p.cout, p.L = B, cin1, cin2, cout, Ln
    p.x, p.x2, p.isc, p.ish, p.in_relu = _p(x), _p(x2), _p(isc), _p(ish), int(in_relu)
    p.wp, p.bias, p.res, p.out_relu = _p(wp), _p(pad32(bias, cout)), _p(res), int(out_relu)
    p.y = _p(y)
    # rows of the statistics partials = workgroups of THIS launch (the library picks the kernel from the block)
    pooled = None
    if pool is not None:
        K, gamma = pool
        S = Ln // K
        ymax = _f32(B, cout, S, device=x.device)
        arg = torch.empty((B, cout, S), dtype=torch.int32, device=x.device)
        p.pool_K, p.pool_gamma, p.pool_ymax, p.pool_arg = K, _p(gamma.detach()), _p(ymax), _p(arg)
        if L.load().pcr_tdense_fwd_pooled(ctypes.byref(p)):
            pooled = (ymax, arg)
        else:
            p.pool_K = 0
    stats = _f32(L.load().pcr_tdense_fwd_groups(ctypes.byref(p)), 2, _c32(cout), device=x.device) if want_stats else None
    p.stats = _p(stats)
    cin = cin1 + cin2
    with _prof("tdense_fwd[cin=%d,cout=%d,L=%d]" % (cin, cout, Ln), 2.0 * B * Ln * cin * cout,
               4.0 * B * Ln * (cin + cout * (2 if res is not None else 1)), arith="lib"):
        L.check(L.load().pcr_tdense_fwd_f32(ctypes.byref(p), L.stream_ptr()), "pcr_tdense_fwd_f32")
    if pool is not None:
        return y, stats, pooled
    return y, stats


def reduce_parts(part, nparts, stride, rows, cols, ld):
    out = _f32(rows, cols, device=part.device)
    L.check(L.load().pcr_reduce_parts_f32(L.ptr(part), nparts, ctypes.c_long(stride), rows, cols, ld, L.ptr(out),
                                          L.stream_ptr()), "pcr_reduce_parts_f32")
    return out


# ---- partial-sum reductions -------------------------------------------------------------------------------------------
# A backward launch leaves per-workgroup partial records (weight / bias / norm gradients side by side); ONE launch
# (pcr_reduce_multi_f32) sums every region of the record into its own compact tensor -- no padded views, so autograd's
# AccumulateGrad takes the gradients as they are instead of cloning them.  (Tried in round 5 and removed: deferring all
# reductions of a backward pass to one launch at its end through an autograd-engine callback.  It measured no faster --
# the cost is the bytes of the partial records, not the ~35 launches -- and it is only safe for a parameter used ONCE in
# the graph: a second use makes the engine add the still-unreduced buffer.)
class _ReduceJob(ctypes.Structure):
    _fields_ = [("part", c_fp), ("out", c_fp), ("stride", ctypes.c_long), ("nparts", ctypes.c_int), ("rows", ctypes.c_int),
                ("cols", ctypes.c_int), ("ld", ctypes.c_int)]


def reduce_regions(part, nparts, stride, regions):
    """regions [(offset in floats, rows, cols, ld)] of every partial record -> compact (rows, cols) tensors, summed over the
    nparts records (increasing record order, fixed: bit-reproducible)"""
    dev = part.device
    outs = [_f32(rows, cols, device=dev) for _, rows, cols, _ in regions]
    jobs = [_ReduceJob(part.data_ptr() + 4 * off, o.data_ptr(), stride, nparts, rows, cols, ld)
            for (off, rows, cols, ld), o in zip(regions, outs)]
    arr = (_ReduceJob * len(jobs))(*jobs)
    L.check(L.load().pcr_reduce_multi_f32(arr, len(jobs), L.stream_ptr()), "pcr_reduce_multi_f32")
    return outs


# Arithmetic of the training launches' matrix phases:
#   "f32"         f32-input MFMA, exact fmaf chains everywhere;
#   "bf16x3"      (default) split bf16 on the bf16 matrix core (three MFMAs per product, f32 accumulation) for the
#                 GRADIENT products only -- dx and dW of the 128 x 128 grouped-MLP layers (pcr_tdense_bwd.precision) and of
#                 the fused attention chains (pcr_attn_tail / pcr_attn_head .precision); every forward value, and with it
#                 every ReLU mask and max-pool winner, is the f32 graph's;
#   "bf16x3_all"  opt-in: the fused chains' forward (and its recomputation) as split bf16 too.  ~0.3 ms per pt128_train
#                 step faster, and ~1 ReLU in 10^5 whose argument lies within 1e-5 of zero opens where the f32 graph keeps
#                 it shut: that token's gradient row moves by per cent, a weight gradient by ~1e-4 of its scale.
# PCR_TRAIN_PRECISION in the environment or set_train_precision() select it.
_TRAIN_PRECISIONS = ("f32", "bf16x3", "bf16x3_all")
TRAIN_PRECISION = os.environ.get("PCR_TRAIN_PRECISION", "bf16x3")
if TRAIN_PRECISION not in _TRAIN_PRECISIONS:
    raise L.PcrError("PCR_TRAIN_PRECISION must be one of %s" % (_TRAIN_PRECISIONS,))


def set_train_precision(name):
    """-> previous setting"""
    global TRAIN_PRECISION
    if name not in _TRAIN_PRECISIONS:
        raise L.PcrError("training precision must be one of %s" % (_TRAIN_PRECISIONS,))
    prev, TRAIN_PRECISION = TRAIN_PRECISION, name
    return prev


def _bwd_bf():
    return TRAIN_PRECISION != "f32"


def pack_both_bf(w):
    """-> (bf16 hi / lo image of W, of W^T): the operands of the fused chains on the bf16 matrix core.  A weight of a
    prepacked model is registered on its first request and refreshed by the one launch per iteration from then on; a
    miss packs on the spot (two small launches)"""
    for t in _PREPACKS.values():
        hit = t.lookup_bf(w)
        if hit is not None:
            return hit
    wd = _dev(w.detach())
    if wd.dim() != 2:                     # (a 1x1 Conv1d / Conv2d weight (d, c, 1[, 1]): its matrix, as _Prepack.build takes it)
        wd = wd.reshape(wd.shape[0], -1)
    rows, cols = wd.shape
    lib = L.load()
    a = _f32(lib.pcr_packed_weight_bf16_floats(rows, cols), device=wd.device)
    b = _f32(lib.pcr_packed_weight_bf16_floats(cols, rows), device=wd.device)
    L.check(lib.pcr_pack_weight_bf16_dev_f32(L.ptr(wd), rows, cols, cols, 0, L.ptr(a), L.stream_ptr()),
            "pcr_pack_weight_bf16_dev_f32")
    L.check(lib.pcr_pack_weight_bf16_dev_f32(L.ptr(wd), rows, cols, cols, 1, L.ptr(b), L.stream_ptr()),
            "pcr_pack_weight_bf16_dev_f32")
    return a, b


def pack_bf_T(w):
    """(rows, cols) device weight -> bf16 hi / lo image of W^T (the dx operand of the bf16 backward), packed on the device:
    one small launch per layer and iteration (the weights change every step)"""
    wd = _dev(w.detach())
    rows, cols = wd.shape
    out = _f32(L.load().pcr_packed_weight_bf16_floats(cols, rows), device=wd.device)
    L.check(L.load().pcr_pack_weight_bf16_dev_f32(L.ptr(wd), rows, cols, cols, 1, L.ptr(out), L.stream_ptr()),
            "pcr_pack_weight_bf16_dev_f32")
    return out


def tdense_bwd(g, x, cout, dy_mode=0, y=None, k=None, argmax=None, pooled=None, K=0, S=0, x2=None, isc=None, ish=None,
               iinv=None, in_relu=False, wpT=None, want_dstats=False, want_dw=True, wpT_bf=None):
    """-> dict(dx, dx2, dstats, dW (cout, cin1+cin2), db (cout)); see pcr_tdense_bwd in include/pcr.h"""
    x = _dev(x)
    B, cin1, Ln = x.shape
    cin2 = 0
    if x2 is not None:
        x2 = _dev(x2)
        cin2 = x2.shape[1]
    cin = cin1 + cin2
    dev = x.device
    out = {}
    p = _TBwd()
    p.B, p.cin1, p.cin2, p.cout, p.L = B, cin1, cin2, cout, Ln
    p.g, p.y, p.dy_mode = _p(g), _p(y), dy_mode
    if k is not None:
        p.ka, p.kb, p.kc = _p(k["ka"]), _p(k["kb"]), _p(k["kc"])
    p.argmax, p.pooled, p.K, p.S = _p(argmax), _p(pooled), K, S
    p.x, p.x2, p.isc, p.ish, p.iinv, p.in_relu = _p(x), _p(x2), _p(isc), _p(ish), _p(iinv), int(in_relu)
    # workgroups of THIS launch = rows of its partial buffers: asked with the wanted outputs marked non-NULL (the
    # library picks the kernel -- and with it the grid -- from the parameter block), then the real buffers go in
    p.wpT = _p(wpT)
    if wpT_bf is not None and _bwd_bf():
        p.wpT_bf, p.precision = _p(wpT_bf), 1
    if wpT is not None:
        p.dx, p.dx2 = 1, (1 if cin2 else None)
        p.dstats = 1 if want_dstats else None
    if want_dw:
        p.dwp = p.dbp = 1
        p.part_stride = _c32(cout) * _c32(cin) + _c32(cout)
    nwg = L.load().pcr_tdense_bwd_groups(ctypes.byref(p))
    p.dx = p.dx2 = p.dstats = p.dwp = p.dbp = None
    if wpT is not None:
        out["dx"] = _f32(B, cin1, Ln, device=dev)
        out["dx2"] = _f32(B, cin2, Ln, device=dev) if cin2 else None
        p.wpT, p.dx, p.dx2 = _p(wpT), _p(out["dx"]), _p(out["dx2"])
        if want_dstats:
            out["dstats"] = _f32(nwg, 2, _c32(cin1), device=dev)
            p.dstats = _p(out["dstats"])
    coutP, cinP = _c32(cout), _c32(cin)
    if want_dw:
        # per workgroup: the dW image (coutP x cinP) followed by db (coutP); ONE reduction over both
        per = coutP * cinP + coutP
        parts = _f32(nwg, per, device=dev)
        p.dwp, p.dbp, p.part_stride = _p(parts), parts.data_ptr() + 4 * coutP * cinP, per
    # algorithmic work: dW = dy f(x)^T and dx = W^T dy (2 B L cout cin flops each); bytes: read g (or the small pooled
    # tensors in mode 3), y (BatchNorm / ReLU modes), x; write dx
    flops = 2.0 * B * Ln * cout * cin * ((1 if want_dw else 0) + (1 if wpT is not None else 0))
    nbytes = 4.0 * B * Ln * ((cout if dy_mode != 3 else 0) + (cout if dy_mode != 0 else 0) + cin +
                             (cin if wpT is not None else 0))
    with _prof("tdense_bwd[mode=%d,cin=%d,cout=%d,L=%d]" % (dy_mode, cin, cout, Ln), flops, nbytes, arith="lib"):
        L.check(L.load().pcr_tdense_bwd_f32(ctypes.byref(p), L.stream_ptr()), "pcr_tdense_bwd_f32")
    if want_dw:
        dW, db = reduce_regions(parts, nwg, per, [(0, cout, cin, cinP), (coutP * cinP, 1, cout, cout)])
        out["dW"], out["db"] = dW, db.view(cout)
    return out


def bn_fwd_finalize(part, nparts, C, R, gamma, beta, eps, momentum, running_mean=None, running_var=None, shift0=None,
                    shift0_stride=1):
    dev = part.device
    o = {k: _f32(C, device=dev) for k in ("scale", "shift", "inv_scale", "mean", "invstd")}
    p = _BnFwd()
    p.part, p.nparts, p.C, p.R = _p(part), nparts, C, float(R)
    p.gamma, p.beta, p.eps, p.momentum = _p(gamma.detach()), _p(beta.detach()), eps, momentum
    p.running_mean, p.running_var = _p(running_mean), _p(running_var)
    p.shift0, p.shift0_stride = _p(shift0), shift0_stride
    for k, v in o.items():
        setattr(p, k, _p(v))
    L.check(L.load().pcr_bn_fwd_finalize_f32(ctypes.byref(p), L.stream_ptr()), "pcr_bn_fwd_finalize_f32")
    return o


def bn_bwd_finalize(part, nparts, C, R, gamma, mean, invstd, centre=None):
    dev = part.device
    o = {k: _f32(C, device=dev) for k in ("ka", "kb", "kc", "dgamma", "dbeta")}
    p = _BnBwd()
    p.part, p.nparts, p.C, p.R = _p(part), nparts, C, float(R)
    p.gamma, p.mean, p.invstd = _p(gamma.detach()), _p(mean), _p(invstd)
    p.centre = _p(centre)
    for k, v in o.items():
        setattr(p, k, _p(v))
    L.check(L.load().pcr_bn_bwd_finalize_f32(ctypes.byref(p), L.stream_ptr()), "pcr_bn_bwd_finalize_f32")
    return o


# ----------------------------------------------------------------------------------- autograd Functions --
class TDense(Function):
    """y = [relu](W [x ; x2] + bias [+ res]) on (B,C,L) tensors; W (cout, cin1+cin2) row-major (nn.Linear / 1x1 conv)"""

    @staticmethod
    def forward(ctx, x, x2, W, bias, res, out_relu):
        cout = W.shape[0]
        need_dx = x.requires_grad or (x2 is not None and x2.requires_grad)
        wp, wpT = pack_both(W) if need_dx else (pack_dev(W), None)
        y, _ = tdense_fwd(x, wp, cout, x2=x2, bias=bias, res=res, out_relu=out_relu)
        ctx.save_for_backward(x, x2, W, y if out_relu else None, wpT)
        ctx.meta = (out_relu, bias is not None, res is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, x2, W, y, wpT = ctx.saved_tensors
        out_relu, has_bias, has_res = ctx.meta
        g = g.contiguous()
        need_dx = ctx.needs_input_grad[0] or (x2 is not None and ctx.needs_input_grad[1])
        if need_dx and wpT is None:
            wpT = pack_dev(W, transpose=True)
        r = tdense_bwd(g, x, W.shape[0], dy_mode=2 if out_relu else 0, y=y, x2=x2,
                       wpT=wpT if need_dx else None,
                       want_dw=ctx.needs_input_grad[2] or (has_bias and ctx.needs_input_grad[3]))
        g_res = None
        if has_res and ctx.needs_input_grad[4]:
            # relu(W x + res): the residual sees the gradient through the same mask (only the wide-layer tiling below
            # builds this combination; one elementwise select)
            g_res = torch.where(y > 0, g, torch.zeros((), dtype=g.dtype, device=g.device)) if out_relu else g
        return (r.get("dx"), r.get("dx2"), r.get("dW"), r.get("db") if has_bias else None, g_res, None)


# pcr_tdense_{fwd,bwd}_f32 take up to 384 output rows and 288 input rows per launch (the backward keeps dy and the forward
# input of a 64-token tile in LDS together).  Wider layers -- the 256-channel attention blocks and 512-row feed-forward /
# table layers of the mul = 2 Point-Transformer -- run as a grid of launches: output rows in chunks, input rows in chunks
# chained through the launch's residual input; autograd sees ordinary Functions, so the backward tiles the same way.
_MAX_COUT, _MAX_CIN, _CHUNK = 384, 288, 256


def dense(x, W, bias=None, x2=None, res=None, relu=False):
    cout, cin = W.shape
    if cout <= _MAX_COUT and cin <= _MAX_CIN and not (relu and res is not None):
        return TDense.apply(x, x2, W, bias, res, relu)
    segs = [x] if x2 is None else [x, x2]             # input = the channel concatenation of the segments
    pieces, c0 = [], 0                                # (tensor, first column of W)
    for t in segs:
        for a in range(0, t.shape[1], _CHUNK):
            b = min(t.shape[1], a + _CHUNK)
            pieces.append((t if (a == 0 and b == t.shape[1]) else t[:, a:b].contiguous(), c0 + a, c0 + b))
        c0 += t.shape[1]
    assert c0 == cin, "dense: input width %d != weight width %d" % (c0, cin)
    outs = []
    for o0 in range(0, cout, _CHUNK):
        o1 = min(cout, o0 + _CHUNK)
        acc = None if res is None else (res if (o0 == 0 and o1 == cout) else res[:, o0:o1].contiguous())
        for i, (t, a, b) in enumerate(pieces):
            last = i == len(pieces) - 1
            acc = TDense.apply(t, None, W[o0:o1, a:b], bias[o0:o1] if (bias is not None and i == 0) else None, acc,
                               relu and last)
        outs.append(acc)
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)


class SaEdgeTrain(Function):
    """Grouped edge MLP of one set-abstraction layer in TRAINING mode: layer 1 from per-point tables, three
    (conv, BatchNorm over the batch, ReLU) layers, max over K.  Inputs: xyz (B,N,3), idx (B,S,K) int32, tab (B,2c1,N)
    or None, then per layer (W, b, gamma, beta); `bns` = the three nn.BatchNorm2d modules (running statistics are
    updated in place, as nn.BatchNorm2d does in training mode).  -> pooled (B,c3,S)."""

    @staticmethod
    def forward(ctx, xyz, idx, tab, wa, b1, g1, be1, w2, b2, g2, be2, w3, b3, g3, be3, bns):
        lib = L.load()
        xyz, idx = _dev(xyz), idx.contiguous()
        L.require_i32(idx)
        B, N, _ = xyz.shape
        _, S, K = idx.shape
        Ln = S * K
        R = B * Ln
        c1, c2, c3 = wa.shape[0], w2.shape[0], w3.shape[0]
        dev = xyz.device
        tab = None if tab is None else _dev(tab)
        y1 = _f32(B, c1, Ln, device=dev)
        st1 = _f32(B, 2, _c32(c1), device=dev)
        with _prof("sa_l1_fwd[c1=%d,N=%d,S=%d,K=%d]" % (c1, N, S, K), 8.0 * B * Ln * c1, 4.0 * B * (Ln * (c1 + 1) + 2 * c1 * N)):
            L.check(lib.pcr_sa_l1_fwd_f32(L.ptr(xyz), L.ptr(idx), L.ptr(tab), L.ptr(_dev(wa.detach())), L.ptr(b1.detach()),
                                          L.ptr(y1), L.ptr(st1), B, N, S, K, c1, L.stream_ptr()), "pcr_sa_l1_fwd_f32")

        def fin(st, nparts, C, gamma, beta, bn):
            if bn.momentum is None:
                # nn.BatchNorm2d(momentum=None) is the cumulative average 1 / num_batches_tracked (a device counter: a
                # host read per layer and step); no reference config uses it -- refused rather than approximated
                raise L.PcrError("BatchNorm with momentum=None (cumulative average) is not supported by the HIP training path")
            o = bn_fwd_finalize(st, nparts, C, R, gamma, beta, bn.eps, bn.momentum,
                                bn.running_mean if bn.track_running_stats else None,
                                bn.running_var if bn.track_running_stats else None)
            if bn.track_running_stats:
                # (written through raw pointers by the finalize launch: keep Tensor._version honest for the caches)
                torch.autograd.graph.increment_version([bn.running_mean, bn.running_var])
                if bn.num_batches_tracked is not None:
                    bn.num_batches_tracked += 1
            return o
        n1 = fin(st1, B, c1, g1, be1, bns[0])
        y2, st2 = tdense_fwd(y1, pack_dev(w2), c2, isc=n1["scale"], ish=n1["shift"], in_relu=True, bias=b2, want_stats=True)
        n2 = fin(st2, st2.shape[0], c2, g2, be2, bns[1])
        # the last layer's launch also finds every centre's winning row where it can (max of the raw output for
        # gamma >= 0, min otherwise: relu(scale y + shift) is monotone in y); the pooled activation is then one affine +
        # ReLU over the (B,c3,S) winners instead of a second pass over the (B,c3,S K) tensor
        y3, st3, won = tdense_fwd(y2, pack_dev(w3), c3, isc=n2["scale"], ish=n2["shift"], in_relu=True, bias=b3,
                                  want_stats=True, pool=(K, g3) if FUSE_POOL else None) if FUSE_POOL else \
            tdense_fwd(y2, pack_dev(w3), c3, isc=n2["scale"], ish=n2["shift"], in_relu=True, bias=b3, want_stats=True) + (None,)
        n3 = fin(st3, st3.shape[0], c3, g3, be3, bns[2])
        pooled = _f32(B, c3, S, device=dev)
        if won is not None:
            ymax, argmax = won
            L.check(lib.pcr_bn_affine_f32(L.ptr(ymax), None, L.ptr(n3["scale"]), L.ptr(n3["shift"]), None, None, None, None,
                                          1, ctypes.c_float(0.0), L.ptr(pooled), B, c3, S, L.stream_ptr()),
                    "pcr_bn_affine_f32")
        else:
            argmax = torch.empty((B, c3, S), dtype=torch.int32, device=dev)
            ymax = _f32(B, c3, S, device=dev)
            with _prof("sa_pool_fwd[c=%d,S=%d,K=%d]" % (c3, S, K), 2.0 * B * Ln * c3, 4.0 * B * c3 * (Ln + 2 * S)):
                L.check(lib.pcr_sa_pool_fwd_f32(L.ptr(y3), L.ptr(n3["scale"]), L.ptr(n3["shift"]), L.ptr(pooled),
                                                L.ptr(argmax), L.ptr(ymax), B, c3, S, K, L.stream_ptr()), "pcr_sa_pool_fwd_f32")
        ctx.save_for_backward(xyz, idx, y1, y2, y3, pooled, argmax, w2, w3, g1, g2, g3, ymax)
        ctx.norms = (n1, n2, n3)
        ctx.has_tab = tab is not None
        ctx.dims = (B, N, S, K, c1, c2, c3)
        ctx.mark_non_differentiable(argmax)
        return pooled

    @staticmethod
    def backward(ctx, gp):
        lib = L.load()
        xyz, idx, y1, y2, y3, pooled, argmax, w2, w3, g1, g2, g3, ymax = ctx.saved_tensors
        n1, n2, n3 = ctx.norms
        B, N, S, K, c1, c2, c3 = ctx.dims
        Ln, R, dev = S * K, B * S * K, xyz.device
        gp = gp.contiguous()
        part3 = _f32(B, 2, _c32(c3), device=dev)
        gz = _f32(B, c3, S, device=dev)          # the pooled gradient where the ReLU is open, zero elsewhere
        L.check(lib.pcr_sa_pool_bwd_stats_f32(L.ptr(gp), L.ptr(pooled), L.ptr(ymax), L.ptr(part3), L.ptr(gz), B, c3, S,
                                              L.stream_ptr()), "pcr_sa_pool_bwd_stats_f32")
        k3 = bn_bwd_finalize(part3, B, c3, R, g3, n3["mean"], n3["invstd"])
        # (128 x 128 layers: dx and dW on the bf16 matrix core when TRAIN_PRECISION says so)
        bf3 = pack_bf_T(w3) if (_bwd_bf() and c3 == 128 and c2 == 128) else None
        bf2 = pack_bf_T(w2) if (_bwd_bf() and c2 == 128 and c1 == 128) else None
        r3 = tdense_bwd(gz, y2, c3, dy_mode=3, y=y3, k=k3, argmax=argmax, pooled=None, K=K, S=S,
                        isc=n2["scale"], ish=n2["shift"], iinv=n2["inv_scale"], in_relu=True,
                        wpT=pack_dev(w3, transpose=True), want_dstats=True, wpT_bf=bf3)
        k2 = bn_bwd_finalize(r3["dstats"], r3["dstats"].shape[0], c2, R, g2, n2["mean"], n2["invstd"])
        r2 = tdense_bwd(r3["dx"], y1, c2, dy_mode=1, y=y2, k=k2, isc=n1["scale"], ish=n1["shift"], iinv=n1["inv_scale"],
                        in_relu=True, wpT=pack_dev(w2, transpose=True), want_dstats=True, wpT_bf=bf2)
        k1 = bn_bwd_finalize(r2["dstats"], r2["dstats"].shape[0], c1, R, g1, n1["mean"], n1["invstd"])
        dtab = _f32(B, 2 * c1, N, device=dev) if ctx.has_tab else None
        dwa_p = _f32(B, c1, 4, device=dev)
        with _prof("sa_l1_bwd[c1=%d,N=%d,S=%d,K=%d]" % (c1, N, S, K), 10.0 * B * Ln * c1, 4.0 * B * (Ln * (2 * c1 + 1) + 2 * c1 * N)):
            L.check(lib.pcr_sa_l1_bwd_f32(L.ptr(xyz), L.ptr(idx), L.ptr(r2["dx"]), L.ptr(y1), L.ptr(k1["ka"]), L.ptr(k1["kb"]),
                                          L.ptr(k1["kc"]), L.ptr(dtab), L.ptr(dwa_p), B, N, S, K, c1, L.stream_ptr()),
                    "pcr_sa_l1_bwd_f32")
        dwa4 = reduce_parts(dwa_p, B, c1 * 4, c1, 4, 4)
        return (None, None, dtab, dwa4[:, :3].contiguous(), dwa4[:, 3].contiguous(), k1["dgamma"], k1["dbeta"],
                r2["dW"], r2["db"], k2["dgamma"], k2["dbeta"], r3["dW"], r3["db"], k3["dgamma"], k3["dbeta"], None)


def sa_edge_train(sa, xyz, feats, idx):
    """PointNetSetAbstractionEdgeSA's grouped MLP + max in training mode: xyz (B,N,3), feats (B,D,N) or None,
    idx (B,S,K) -> (B,c3,S).  The first conv's weight [Wa | Wc | Wf] is split here; autograd carries the table
    weights' gradient back into it."""
    convs, bns = list(sa.mlp_convs), list(sa.mlp_bns)
    c1 = convs[0].weight.shape[0]
    w1 = convs[0].weight.view(c1, -1)
    wa = w1[:, :3]
    tab = None
    if feats is not None:
        D = feats.shape[1]
        wc, wf = w1[:, 3:3 + D], w1[:, 3 + D:3 + 2 * D]
        tab = dense(feats, torch.cat([wf, wc - wf], dim=0))          # (B, 2 c1, N): [P ; Q]
    args = [xyz, idx, tab, wa, convs[0].bias, bns[0].weight, bns[0].bias]
    for l in (1, 2):
        args += [convs[l].weight.view(convs[l].weight.shape[0], -1), convs[l].bias, bns[l].weight, bns[l].bias]
    return SaEdgeTrain.apply(*args, bns)


# ---------------------------------------------------------------- attention / norm / pooling Functions --
class _LinAttnP(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("Lq", ctypes.c_int), ("Sk", ctypes.c_int), ("d", ctypes.c_int), ("H", ctypes.c_int),
                ("eps", ctypes.c_float), ("q", c_fp), ("k", c_fp), ("v", c_fp),
                ("q_bs", ctypes.c_long), ("k_bs", ctypes.c_long), ("v_bs", ctypes.c_long),
                ("out", c_fp), ("A", c_fp), ("ks", c_fp), ("dout", c_fp), ("dq", c_fp), ("dk", c_fp), ("dv", c_fp),
                ("dq_bs", ctypes.c_long), ("dk_bs", ctypes.c_long), ("dv_bs", ctypes.c_long), ("kv_roll", ctypes.c_int)]


def _block(t, d):
    """(pointer, batch stride) of a (B,d,L) channel-major block that may be a channel slice of a wider tensor"""
    assert t.dim() == 3 and t.shape[1] == d and t.stride(2) == 1 and t.stride(1) == t.shape[2], \
        "attention operands must be channel slices of contiguous (B,C,L) tensors"
    return t.data_ptr(), t.stride(0)


def _linattn_fwd(q, k, v, H, eps, kv_roll=0):
    L.require_cuda(q, k, v)
    B, d, Lq = q.shape
    Sk = k.shape[2]
    dev = q.device
    out = _f32(B, d, Lq, device=dev)
    A = _f32(B, H, d // H, d // H, device=dev)
    ks = _f32(B, H, d // H, device=dev)
    p = _LinAttnP()
    p.B, p.Lq, p.Sk, p.d, p.H, p.eps, p.kv_roll = B, Lq, Sk, d, H, eps, kv_roll
    (p.q, p.q_bs), (p.k, p.k_bs), (p.v, p.v_bs) = _block(q, d), _block(k, d), _block(v, d)
    p.out, p.A, p.ks = _p(out), _p(A), _p(ks)
    L.check(L.load().pcr_linattn_fwd_f32(ctypes.byref(p), L.stream_ptr()), "pcr_linattn_fwd_f32")
    return out, A, ks


def _linattn_bwd(q, k, v, A, ks, g, dq, dk, dv, H, eps, kv_roll=0):
    B, d, Lq = q.shape
    p = _LinAttnP()
    p.B, p.Lq, p.Sk, p.d, p.H, p.eps, p.kv_roll = B, Lq, k.shape[2], d, H, eps, kv_roll
    (p.q, p.q_bs), (p.k, p.k_bs), (p.v, p.v_bs) = _block(q, d), _block(k, d), _block(v, d)
    p.A, p.ks, p.dout = _p(A), _p(ks), _p(g)
    (p.dq, p.dq_bs), (p.dk, p.dk_bs), (p.dv, p.dv_bs) = _block(dq, d), _block(dk, d), _block(dv, d)
    L.check(L.load().pcr_linattn_bwd_f32(ctypes.byref(p), L.stream_ptr()), "pcr_linattn_bwd_f32")


class LinAttn(Function):
    """LinearAttention (pointnet2_utils.py:26-47) on channel-major tensors: q (B,d,Lq), k, v (B,d,Sk) -> (B,d,Lq)"""

    @staticmethod
    def forward(ctx, q, k, v, H, eps):
        out, A, ks = _linattn_fwd(q, k, v, H, eps)
        ctx.save_for_backward(q, k, v, A, ks)
        ctx.meta = (H, eps)
        return out

    @staticmethod
    def backward(ctx, g):
        q, k, v, A, ks = ctx.saved_tensors
        H, eps = ctx.meta
        dq, dk, dv = torch.empty_like(q, memory_format=torch.contiguous_format), \
            torch.empty_like(k, memory_format=torch.contiguous_format), torch.empty_like(v, memory_format=torch.contiguous_format)
        _linattn_bwd(q, k, v, A, ks, g.contiguous(), dq, dk, dv, H, eps)
        return dq, dk, dv, None, None


class LinAttnQKV(Function):
    """the same on ONE fused projection qkv (B,3d,L) = [q ; k ; v] (self-attention): the kernels address the three
    channel blocks by a batch stride, and the gradient comes back as one (B,3d,L) buffer -- no slice / pad / add nodes
    in the autograd graph"""

    @staticmethod
    def forward(ctx, qkv, H, eps, kv_roll=0):
        """kv_roll: query cloud b attends to the keys / values of cloud (b + kv_roll) % B of the same buffer (the matching
        stages: every cloud against its pair partner, ReIDNet.py:231-247) -- no rolled copy of the batch"""
        qkv = _dev(qkv)
        d = qkv.shape[1] // 3
        out, A, ks = _linattn_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], H, eps, kv_roll)
        ctx.save_for_backward(qkv, A, ks)
        ctx.meta = (H, eps, kv_roll)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, A, ks = ctx.saved_tensors
        H, eps, kv_roll = ctx.meta
        d = qkv.shape[1] // 3
        buf = torch.empty_like(qkv)
        _linattn_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], A, ks, g.contiguous(),
                     buf[:, :d], buf[:, d:2 * d], buf[:, 2 * d:], H, eps, kv_roll)
        return buf, None, None, None


class LocalAttn(Function):
    """core of local_self_attention in training mode (attention.py:262-296): qkv (B,3C,N) = the fused q | k | v
    projection per POINT, idx (B,N,K) int32 feature-space neighbours (no gradient) -> msg (B,C,N); the backward's
    per-edge gradients are folded onto the points by the grouping backward (owner-computes, no float atomics)"""

    @staticmethod
    def forward(ctx, qkv, idx, H, eps):
        qkv, idx = _dev(qkv), idx.contiguous()
        L.require_i32(idx)
        B, C3, N = qkv.shape
        C, K = C3 // 3, idx.shape[2]
        msg = _f32(B, C, N, device=qkv.device)
        L.check(L.load().pcr_local_attn_train_fwd_f32(L.ptr(qkv), L.ptr(idx), L.ptr(msg), B, N, C, K, H,
                                                      ctypes.c_float(eps), L.stream_ptr()), "pcr_local_attn_train_fwd_f32")
        ctx.save_for_backward(qkv, idx)
        ctx.meta = (H, eps)
        return msg

    @staticmethod
    def backward(ctx, g):
        qkv, idx = ctx.saved_tensors
        H, eps = ctx.meta
        B, C3, N = qkv.shape
        C, K = C3 // 3, idx.shape[2]
        lib = L.load()
        g = g.contiguous()
        dqkv = torch.zeros((B, 3 * C, N), dtype=torch.float32, device=qkv.device)      # (k | v rows: accumulated into)
        edge = _f32(B, 2 * C, N, K, device=qkv.device)
        L.check(lib.pcr_local_attn_train_bwd_f32(L.ptr(qkv), L.ptr(idx), L.ptr(g), L.ptr(dqkv), ctypes.c_long(3 * C * N),
                                                 L.ptr(edge), B, N, C, K, H, ctypes.c_float(eps), L.stream_ptr()),
                "pcr_local_attn_train_bwd_f32")
        dkv = torch.zeros((B, 2 * C, N), dtype=torch.float32, device=qkv.device)
        L.check(lib.pcr_group_bwd_f32(L.ptr(edge), L.ptr(idx), L.ptr(dkv), B, 2 * C, N, N, K, L.stream_ptr()),
                "pcr_group_bwd_f32")
        dqkv[:, C:] = dkv
        return dqkv, None, None, None


class TNorm(Function):
    """[relu](LayerNorm (G = 1) / GroupNorm over the channels of every token of (B,C,L) [+ res])"""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, G, eps, relu):
        x = _dev(x)
        B, C, Ln = x.shape
        y = torch.empty_like(x)
        mean, rstd = _f32(B, G, Ln, device=x.device), _f32(B, G, Ln, device=x.device)
        L.check(L.load().pcr_tnorm_fwd_f32(L.ptr(x), L.ptr(gamma.detach()), L.ptr(beta.detach()),
                                           L.ptr(None if res is None else _dev(res)), L.ptr(y), L.ptr(mean), L.ptr(rstd),
                                           B, C, Ln, G, ctypes.c_float(eps), int(relu), L.stream_ptr()), "pcr_tnorm_fwd_f32")
        ctx.save_for_backward(x, gamma, mean, rstd, y if relu else None)
        ctx.meta = (G, res is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, gamma, mean, rstd, y = ctx.saved_tensors
        G, has_res = ctx.meta
        B, C, Ln = x.shape
        g = g.contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if (has_res and y is not None) else None
        nparts = (B * Ln + 63) // 64              # one partial row per 64-token wave
        part = _f32(nparts, 2, C, device=x.device)
        L.check(L.load().pcr_tnorm_bwd_f32(L.ptr(g), L.ptr(x), L.ptr(gamma.detach()), L.ptr(mean), L.ptr(rstd), L.ptr(y),
                                           L.ptr(dx), L.ptr(dres), L.ptr(part), B, C, Ln, G, L.stream_ptr()),
                "pcr_tnorm_bwd_f32")
        dgam, dbet = reduce_regions(part, nparts, 2 * C, [(0, 1, C, C), (C, 1, C, C)])
        return dx, dgam.view(C), dbet.view(C), (dres if dres is not None else g) if has_res else None, None, None, None


def tnorm(x, norm, res=None, relu=False):
    L.require_default_eps(norm)
    G = getattr(norm, "num_groups", 1)
    return TNorm.apply(x, norm.weight, norm.bias, res, G, norm.eps, relu)


# ------------------------------------------------------------------------------- fused per-token chains --
# PCR_TRAIN_FUSED=0 keeps the unfused graph (one launch per layer: the form of rounds 2-4, and the yardstick of
# tests/test_gpu_train_chain.py)
FUSED_CHAINS = os.environ.get("PCR_TRAIN_FUSED", "1") != "0"


class _AttnTailP(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("L", ctypes.c_int), ("d", ctypes.c_int), ("c1", ctypes.c_int), ("hid", ctypes.c_int),
                ("out", ctypes.c_int), ("residual", ctypes.c_int), ("eps", ctypes.c_float),
                ("msg", c_fp), ("res", c_fp), ("wm", c_fp), ("w0", c_fp), ("w2", c_fp), ("wmT", c_fp), ("w0T", c_fp),
                ("w2T", c_fp), ("g1", c_fp), ("b1", c_fp), ("g2", c_fp), ("b2", c_fp), ("outp", c_fp), ("dout", c_fp),
                ("dmsg", c_fp), ("dres", c_fp), ("parts", c_fp), ("part_stride", ctypes.c_long),
                ("precision", ctypes.c_int), ("fwd_precision", ctypes.c_int)]


def _chain_bf():
    """-> (backward matrix phases on the bf16 core, forward chain too)"""
    return TRAIN_PRECISION != "f32", TRAIN_PRECISION == "bf16x3_all"


def _chain_images(w, bf):
    """(image of W for the forward chain, image of W^T for the backward) in the formats the arithmetic `bf` = (bwd, fwd)
    reads: bf16 hi / lo images where that side runs on the bf16 matrix core, pcr_pack_weight images where it does not"""
    f32 = pack_both(w) if not (bf[0] and bf[1]) else None
    b16 = pack_both_bf(w) if (bf[0] or bf[1]) else None
    return (b16[0] if bf[1] else f32[0]), (b16[1] if (bf[0] or bf[1]) else f32[1])


def _tail_params(msg, res, Wm, g1, b1, W0, W2, g2, b2, residual, eps, images, bf):
    B, d, Ln = msg.shape
    c1, hid, out = res.shape[1], W0.shape[0], W2.shape[0]
    p = _AttnTailP()
    p.B, p.L, p.d, p.c1, p.hid, p.out, p.residual, p.eps = B, Ln, d, c1, hid, out, int(residual), eps
    p.msg, p.res = _p(msg), _p(res)
    (p.wm, p.wmT), (p.w0, p.w0T), (p.w2, p.w2T) = [(_p(a), _p(b)) for a, b in images]
    p.g1, p.b1, p.g2, p.b2 = _p(g1.detach()), _p(b1.detach()), _p(g2.detach()), _p(b2.detach())
    p.precision, p.fwd_precision = int(bf[0] or bf[1]), int(bf[1])
    return p, (B, d, Ln, c1, hid, out)


class AttnTail(Function):
    """the tail of an attention block as ONE launch each way (pcr_attn_tail_{fwd,bwd}_f32):
    out = LN2(W2 relu(W0 [res ; LN1(Wm msg)])) [+ res]; nothing but msg and res is kept for the backward"""

    @staticmethod
    def forward(ctx, msg, res, Wm, g1, b1, W0, W2, g2, b2, residual, eps):
        msg, res = _dev(msg), _dev(res)
        bf = _chain_bf()
        images = [_chain_images(Wm, bf), _chain_images(W0, bf), _chain_images(W2, bf)]
        p, (B, d, Ln, c1, hid, out) = _tail_params(msg, res, Wm, g1, b1, W0, W2, g2, b2, residual, eps, images, bf)
        y = _f32(B, out, Ln, device=msg.device)
        p.outp = _p(y)
        flops = 2.0 * B * Ln * (d * d + (c1 + d) * hid + hid * out)
        with _prof("attn_tail_fwd[d=%d,c1=%d,hid=%d,out=%d,L=%d]" % (d, c1, hid, out, Ln), flops,
                   4.0 * B * Ln * (d + c1 + out), arith="lib"):
            L.check(L.load().pcr_attn_tail_fwd_f32(ctypes.byref(p), L.stream_ptr()), "pcr_attn_tail_fwd_f32")
        ctx.save_for_backward(msg, res, Wm, g1, b1, W0, W2, g2, b2, *[t for im in images for t in im])
        ctx.meta = (residual, eps, bf)
        return y

    @staticmethod
    def backward(ctx, g):
        msg, res, Wm, g1, b1, W0, W2, g2, b2, *flat = ctx.saved_tensors
        residual, eps, bf = ctx.meta
        images = [(flat[0], flat[1]), (flat[2], flat[3]), (flat[4], flat[5])]
        p, (B, d, Ln, c1, hid, out) = _tail_params(msg, res, Wm, g1, b1, W0, W2, g2, b2, residual, eps, images, bf)
        lib = L.load()
        g = _dev(g)
        dev = msg.device
        dmsg, dres = torch.empty_like(msg), torch.empty_like(res)
        rec = lib.pcr_attn_tail_part_floats(d, c1, hid, out)
        nwg = lib.pcr_attn_tail_groups(ctypes.byref(p))
        parts = _f32(nwg, rec, device=dev)
        p.dout, p.dmsg, p.dres, p.parts, p.part_stride = _p(g), _p(dmsg), _p(dres), _p(parts), rec
        flops = 2.0 * B * Ln * (d * d + (c1 + d) * hid + hid * out)
        with _prof("attn_tail_bwd[d=%d,c1=%d,hid=%d,out=%d,L=%d]" % (d, c1, hid, out, Ln), 3.0 * flops,
                   4.0 * B * Ln * (2 * d + 2 * c1 + out), arith="lib"):
            L.check(lib.pcr_attn_tail_bwd_f32(ctypes.byref(p), L.stream_ptr()), "pcr_attn_tail_bwd_f32")
        cup = _c32(c1 + d)
        o0, o2 = d * d, d * d + hid * cup
        og = o2 + out * hid
        dWm, dW0, dW2, dg1, db1, dg2, db2 = reduce_regions(
            parts, nwg, rec, [(0, d, d, d), (o0, hid, c1 + d, cup), (o2, out, hid, hid), (og, 1, d, d), (og + d, 1, d, d),
                              (og + 2 * d, 1, out, out), (og + 2 * d + out, 1, out, out)])
        return dmsg, dres, dWm, dg1.view(d), db1.view(d), dW0, dW2, dg2.view(out), db2.view(out), None, None


def attn_tail(m, msg, res, residual, names=("merge", "norm1", "mlp", "norm2")):
    """fused tail of attention block `m` (merge / norm1 / mlp[0], mlp[2] / norm2; `names` for local_self_attention's
    *_knn members), or None when the shape has no fused instantiation (the caller then runs the unfused graph)"""
    if not FUSED_CHAINS:
        return None
    merge, n1, mlp, n2 = (getattr(m, k) for k in names)
    d, c1 = msg.shape[1], res.shape[1]
    hid, out = mlp[0].weight.shape[0], mlp[2].weight.shape[0]
    if (merge.weight.shape != (d, d) or mlp[0].weight.shape[1] != c1 + d or mlp[2].weight.shape[1] != hid or
            getattr(n1, "num_groups", 1) != 1 or getattr(n2, "num_groups", 1) != 1 or
            n1.weight.numel() != d or n2.weight.numel() != out or n1.eps != n2.eps or
            not L.load().pcr_attn_tail_ok(d, c1, hid, out, int(bool(residual)))):
        return None
    L.require_default_eps(n1)
    return AttnTail.apply(msg, res, merge.weight, n1.weight, n1.bias, mlp[0].weight, mlp[2].weight, n2.weight, n2.bias,
                          bool(residual), float(n1.eps))


class _AttnHeadP(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int), ("L", ctypes.c_int), ("c", ctypes.c_int), ("hd", ctypes.c_int), ("d", ctypes.c_int),
                ("np", ctypes.c_int), ("src", ctypes.c_int), ("x", c_fp), ("xyz", c_fp), ("p1", c_fp), ("p2", c_fp),
                ("c1", c_fp), ("c2", c_fp), ("p2T", c_fp), ("w", c_fp * 3), ("wT", c_fp * 3), ("outp", c_fp),
                ("dout", c_fp), ("dx", c_fp), ("parts", c_fp), ("part_stride", ctypes.c_long),
                ("precision", ctypes.c_int), ("fwd_precision", ctypes.c_int)]


def _head_params(x, xyz, P1, c1, P2, c2, src, Ws, images, bf):
    B, C, Ln = x.shape
    hd, d, n = P1.shape[0], Ws[0].shape[0], len(Ws)
    p = _AttnHeadP()
    p.B, p.L, p.c, p.hd, p.d, p.np, p.src = B, Ln, C, hd, d, n, src
    p.x, p.xyz = _p(x), _p(xyz)
    p.p1, p.p2, p.p2T = _p(images[0][0]), _p(images[1][0]), _p(images[1][1])
    p.c1, p.c2 = _p(pad32(c1, hd)), _p(pad32(c2, C))
    for j in range(n):
        p.w[j], p.wT[j] = _p(images[2 + j][0]), _p(images[2 + j][1])
    p.precision, p.fwd_precision = int(bf[0] or bf[1]), int(bf[1])
    return p, (B, C, Ln, hd, d, n)


class AttnHead(Function):
    """the head of an attention block as ONE launch each way (pcr_attn_head_{fwd,bwd}_f32): fp = x + P2 relu(P1 xyz + c1)
    + c2, out (B, n d, L) = [W_j s_j] with s_j = fp (bit j of src) or x; nothing but x and xyz is kept for the backward"""

    @staticmethod
    def forward(ctx, x, xyz, P1, c1, P2, c2, src, *Ws):
        x, xyz = _dev(x), _dev(xyz)
        bf = _chain_bf()                                  # (P1: three input channels, an f32 image either way)
        images = [pack_both(P1), _chain_images(P2, bf)] + [_chain_images(W, bf) for W in Ws]
        p, (B, C, Ln, hd, d, n) = _head_params(x, xyz, P1, c1, P2, c2, src, Ws, images, bf)
        out = _f32(B, n * d, Ln, device=x.device)
        p.outp = _p(out)
        flops = 2.0 * B * Ln * (3 * hd + hd * C + n * d * C)
        with _prof("attn_head_fwd[c=%d,hd=%d,d=%d,n=%d,L=%d]" % (C, hd, d, n, Ln), flops, 4.0 * B * Ln * (C + 3 + n * d),
                   arith="lib"):
            L.check(L.load().pcr_attn_head_fwd_f32(ctypes.byref(p), L.stream_ptr()), "pcr_attn_head_fwd_f32")
        ctx.save_for_backward(x, xyz, P1, c1, P2, c2, *Ws, *[t for im in images for t in im])
        ctx.meta = (src, len(Ws), bf)
        return out

    @staticmethod
    def backward(ctx, g):
        src, n, bf = ctx.meta
        x, xyz, P1, c1, P2, c2, *rest = ctx.saved_tensors
        Ws, flat = rest[:n], rest[n:]
        images = [(flat[2 * i], flat[2 * i + 1]) for i in range(2 + n)]
        p, (B, C, Ln, hd, d, n) = _head_params(x, xyz, P1, c1, P2, c2, src, Ws, images, bf)
        lib = L.load()
        g = _dev(g)
        dx = torch.empty_like(x)
        rec = lib.pcr_attn_head_part_floats(C, hd, d, n, src)
        nwg = lib.pcr_attn_head_groups(ctypes.byref(p))
        parts = _f32(nwg, rec, device=x.device)
        p.dout, p.dx, p.parts, p.part_stride = _p(g), _p(dx), _p(parts), rec
        flops = 2.0 * B * Ln * (3 * hd + hd * C + n * d * C)
        with _prof("attn_head_bwd[c=%d,hd=%d,d=%d,n=%d,L=%d]" % (C, hd, d, n, Ln), 3.0 * flops,
                   4.0 * B * Ln * (2 * C + 3 + n * d), arith="lib"):
            L.check(lib.pcr_attn_head_bwd_f32(ctypes.byref(p), L.stream_ptr()), "pcr_attn_head_bwd_f32")
        o_p2 = hd * 32
        o_w = o_p2 + C * hd
        o_c1 = o_w + n * d * C
        regs = [(0, hd, 3, 32), (o_c1, 1, hd, hd), (o_p2, C, hd, hd), (o_c1 + hd, 1, C, C)] + \
               [(o_w + j * d * C, d, C, C) for j in range(n)]
        dP1, dc1, dP2, dc2, *dWs = reduce_regions(parts, nwg, rec, regs)
        return (dx, None, dP1, dc1.view(hd), dP2, dc2.view(C), None, *dWs)


def attn_head(pos_mlp, x, xyz_cm, Ws, src):
    """fused head: position MLP `pos_mlp` = [Linear(3, hd), ReLU, Linear(hd, c)] added to x, then the projections Ws
    ((d, c) weights; bit j of src: projection j reads x + position code, else x) -> (B, len(Ws) d, L), or None when the
    shape has no fused instantiation"""
    if not FUSED_CHAINS:
        return None
    p1, p2 = pos_mlp[0], pos_mlp[2]
    C, d = x.shape[1], Ws[0].shape[0]
    if (p1.weight.shape[1] != 3 or p2.weight.shape != (C, p1.weight.shape[0]) or p1.bias is None or p2.bias is None or
            any(W.shape != (d, C) for W in Ws) or xyz_cm.shape[1] != 3 or xyz_cm.shape[2] != x.shape[2] or
            not L.load().pcr_attn_head_ok(C, p1.weight.shape[0], d, len(Ws), src)):
        return None
    return AttnHead.apply(x, xyz_cm, p1.weight, p1.bias, p2.weight, p2.bias, src, *Ws)


class LinAttnKV(Function):
    """LinAttn with k | v as the two channel halves of ONE (B,2d,Sk) tensor (the fused head's output): the gradient comes
    back as one buffer -- no slice / pad / add nodes in the autograd graph"""

    @staticmethod
    def forward(ctx, q, kv, H, eps):
        q, kv = _dev(q), _dev(kv)
        d = q.shape[1]
        out, A, ks = _linattn_fwd(q, kv[:, :d], kv[:, d:], H, eps)
        ctx.save_for_backward(q, kv, A, ks)
        ctx.meta = (H, eps)
        return out

    @staticmethod
    def backward(ctx, g):
        q, kv, A, ks = ctx.saved_tensors
        H, eps = ctx.meta
        d = q.shape[1]
        dq, buf = torch.empty_like(q), torch.empty_like(kv)
        _linattn_bwd(q, kv[:, :d], kv[:, d:], A, ks, g.contiguous(), dq, buf[:, :d], buf[:, d:], H, eps)
        return dq, buf, None, None


class PoolPair(Function):
    """o (2P,C,L) -> (P,2C): [max, mean] over the point-concatenated pair (clouds p and p + P)"""

    @staticmethod
    def forward(ctx, o):
        o = _dev(o)
        twoP, C, Ln = o.shape
        P = twoP // 2
        pooled = _f32(P, 2 * C, device=o.device)
        arg = torch.empty((P, C), dtype=torch.int32, device=o.device)
        L.check(L.load().pcr_pool_pair_fwd_f32(L.ptr(o), L.ptr(pooled), L.ptr(arg), P, C, Ln, L.stream_ptr()),
                "pcr_pool_pair_fwd_f32")
        ctx.save_for_backward(arg)
        ctx.dims = (P, C, Ln)
        return pooled

    @staticmethod
    def backward(ctx, g):
        arg, = ctx.saved_tensors
        P, C, Ln = ctx.dims
        g = g.contiguous()
        dout = _f32(2 * P, C, Ln, device=g.device)
        L.check(L.load().pcr_pool_pair_bwd_f32(L.ptr(g), L.ptr(arg), L.ptr(dout), P, C, Ln, L.stream_ptr()),
                "pcr_pool_pair_bwd_f32")
        return dout


class PoolBoth(Function):
    """o (P,C,L) -> (P,2C): [max over L, mean over L] (get_pooled_feats 'both' on one tensor, ReIDNet.py:529-532)"""

    @staticmethod
    def forward(ctx, o):
        o = _dev(o)
        P, C, Ln = o.shape
        pooled = _f32(P, 2 * C, device=o.device)
        arg = torch.empty((P, C), dtype=torch.int32, device=o.device)
        L.check(L.load().pcr_pool_both_fwd_f32(L.ptr(o), L.ptr(pooled), L.ptr(arg), P, C, Ln, L.stream_ptr()),
                "pcr_pool_both_fwd_f32")
        ctx.save_for_backward(arg)
        ctx.dims = (P, C, Ln)
        return pooled

    @staticmethod
    def backward(ctx, g):
        arg, = ctx.saved_tensors
        P, C, Ln = ctx.dims
        g = g.contiguous()
        dout = _f32(P, C, Ln, device=g.device)
        L.check(L.load().pcr_pool_both_bwd_f32(L.ptr(g), L.ptr(arg), L.ptr(dout), P, C, Ln, L.stream_ptr()),
                "pcr_pool_both_bwd_f32")
        return dout


class ChannelMax(Function):
    """x (B,C,L) -> (B,C/W,L): max over windows of W channels of every point (get_pooled_feats 'max', ReIDNet.py:145,
    526-528: nn.MaxPool1d(W) on the permuted tensor)"""

    @staticmethod
    def forward(ctx, x, W):
        x = _dev(x)
        B, C, Ln = x.shape
        y = _f32(B, C // W, Ln, device=x.device)
        arg = torch.empty((B, C // W, Ln), dtype=torch.int32, device=x.device)
        L.check(L.load().pcr_channel_max_fwd_f32(L.ptr(x), L.ptr(y), L.ptr(arg), B, C, Ln, W, L.stream_ptr()),
                "pcr_channel_max_fwd_f32")
        ctx.save_for_backward(arg)
        ctx.dims = (B, C, Ln, W)
        return y

    @staticmethod
    def backward(ctx, g):
        arg, = ctx.saved_tensors
        B, C, Ln, W = ctx.dims
        g = g.contiguous()
        dx = _f32(B, C, Ln, device=g.device)
        L.check(L.load().pcr_channel_max_bwd_f32(L.ptr(g), L.ptr(arg), L.ptr(dx), B, C, Ln, W, L.stream_ptr()),
                "pcr_channel_max_bwd_f32")
        return dx, None


# --------------------------------------------------------------- PointNet pieces (csrc/train_bn_kernels.hip) --
class BnAct(Function):
    """act(BatchNorm(y)) with BATCH statistics over (B, L) per channel on a (B,C,L) tensor; act: None, "relu" or
    ("leaky", slope).  Running statistics are updated in place as nn.BatchNorm1d does in training mode.  (The
    Point-Transformer path folds its BatchNorms into the train-dense launches; PointNet's and DGCNN's stand between
    layers that cannot take them.)"""

    @staticmethod
    def forward(ctx, y, gamma, beta, bn, act, slope):
        y = _dev(y)
        B, C, Ln = y.shape
        lib = L.load()
        if bn.momentum is None:
            raise L.PcrError("BatchNorm with momentum=None (cumulative average) is not supported by the HIP training path")
        nparts = max(1, min(B, 2048 // max(C, 1)))
        part = _f32(nparts, 2, _c32(C), device=y.device)
        # sums of y - y[0][c][0] (first element of the channel as the offset): see pcr_bn_fwd_fin.shift0
        L.check(lib.pcr_bn_sums_f32(L.ptr(y), None, None, None, 0, ctypes.c_float(0.0), None, 1, L.ptr(part), nparts, B, C, Ln,
                                    L.stream_ptr()), "pcr_bn_sums_f32")
        n = bn_fwd_finalize(part, nparts, C, B * Ln, gamma, beta, bn.eps, bn.momentum,
                            bn.running_mean if bn.track_running_stats else None,
                            bn.running_var if bn.track_running_stats else None, shift0=y, shift0_stride=Ln)
        if bn.track_running_stats:
            torch.autograd.graph.increment_version([bn.running_mean, bn.running_var])
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
        z = _f32(B, C, Ln, device=y.device)
        L.check(lib.pcr_bn_affine_f32(L.ptr(y), None, L.ptr(n["scale"]), L.ptr(n["shift"]), None, None, None, None,
                                      int(act), ctypes.c_float(slope), L.ptr(z), B, C, Ln, L.stream_ptr()),
                "pcr_bn_affine_f32")
        ctx.save_for_backward(y, gamma)
        ctx.norm, ctx.act, ctx.slope, ctx.nparts = n, int(act), float(slope), nparts
        return z

    @staticmethod
    def backward(ctx, g):
        y, gamma = ctx.saved_tensors
        n, act, slope, nparts = ctx.norm, ctx.act, ctx.slope, ctx.nparts
        B, C, Ln = y.shape
        g = g.contiguous()
        lib = L.load()
        part = _f32(nparts, 2, _c32(C), device=y.device)
        L.check(lib.pcr_bn_sums_f32(L.ptr(y), L.ptr(g), L.ptr(n["scale"]), L.ptr(n["shift"]), act, ctypes.c_float(slope),
                                    L.ptr(n["mean"]), 0, L.ptr(part), nparts, B, C, Ln, L.stream_ptr()), "pcr_bn_sums_f32")
        k = bn_bwd_finalize(part, nparts, C, B * Ln, gamma, n["mean"], n["invstd"], centre=n["mean"])
        dy = _f32(B, C, Ln, device=y.device)
        kc = k["ka"] * k["dbeta"] * (-1.0 / (B * Ln))        # centred form: dy = ka g' + kb (y - mean) - ka dbeta / R
        L.check(lib.pcr_bn_affine_f32(L.ptr(y), L.ptr(g), L.ptr(k["ka"]), L.ptr(k["kb"]), L.ptr(kc), L.ptr(n["scale"]),
                                      L.ptr(n["shift"]), L.ptr(n["mean"]), act, ctypes.c_float(slope), L.ptr(dy), B, C, Ln,
                                      L.stream_ptr()), "pcr_bn_affine_f32")
        return dy, k["dgamma"], k["dbeta"], None, None, None


def bn_act(y, bn, relu, slope=0.0):
    """relu: apply the activation z > 0 ? z : slope z after the norm (slope 0: ReLU)"""
    return BnAct.apply(y, bn.weight, bn.bias, bn, bool(relu), slope)


class EdgeConvTrain(Function):
    """One EdgeConv layer of DGCNN in training mode (dgcnn_orig.py:32-56, 127-143): Conv2d(2C -> Co, no bias) on
    [f_j - f_i ; f_i] + BatchNorm2d over all B N k edges (batch statistics) + LeakyReLU + max over the k neighbours.
    The conv is decomposed as in the inference path, W [f_j - f_i ; f_i] = W1 f_j + (W2 - W1) f_i: `tab` (B,2Co,N) =
    [W1 f ; (W2 - W1) f] comes from ONE train-dense launch (autograd carries the table gradient back into the weight),
    and the per-edge pre-activation y = tab[:Co][idx] + tab[Co:][i] is the first layer of the grouped set-abstraction
    MLP with no coordinate term -- pcr_sa_l1_{fwd,bwd}_f32 with the points as their own centres."""

    @staticmethod
    def forward(ctx, tab, idx, gamma, beta, bn, slope):
        lib = L.load()
        tab, idx = _dev(tab), idx.contiguous()
        L.require_i32(idx)
        B, two_co, N = tab.shape
        Co = two_co // 2
        K = idx.shape[2]
        dev = tab.device
        if bn.momentum is None:
            raise L.PcrError("BatchNorm with momentum=None (cumulative average) is not supported by the HIP training path")
        xyz0 = torch.zeros((B, N, 3), dtype=torch.float32, device=dev)       # (no coordinate term: zero weights below)
        wa0 = torch.zeros((Co, 3), dtype=torch.float32, device=dev)
        b0 = torch.zeros((Co,), dtype=torch.float32, device=dev)
        y = _f32(B, Co, N * K, device=dev)
        st = _f32(B, 2, _c32(Co), device=dev)
        L.check(lib.pcr_sa_l1_fwd_f32(L.ptr(xyz0), L.ptr(idx), L.ptr(tab), L.ptr(wa0), L.ptr(b0), L.ptr(y), L.ptr(st),
                                      B, N, N, K, Co, L.stream_ptr()), "pcr_sa_l1_fwd_f32")
        R = B * N * K
        n = bn_fwd_finalize(st, B, Co, R, gamma, beta, bn.eps, bn.momentum,
                            bn.running_mean if bn.track_running_stats else None,
                            bn.running_var if bn.track_running_stats else None)
        if bn.track_running_stats:
            torch.autograd.graph.increment_version([bn.running_mean, bn.running_var])
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked += 1
        pooled = _f32(B, Co, N, device=dev)
        arg = torch.empty((B, Co, N), dtype=torch.int32, device=dev)
        yraw = _f32(B, Co, N, device=dev)
        L.check(lib.pcr_edge_pool_fwd_f32(L.ptr(y), L.ptr(n["scale"]), L.ptr(n["shift"]), ctypes.c_float(slope), L.ptr(pooled),
                                          L.ptr(arg), L.ptr(yraw), B, Co, N, K, L.stream_ptr()), "pcr_edge_pool_fwd_f32")
        ctx.save_for_backward(idx, y, pooled, arg, yraw, gamma, xyz0)
        ctx.norm, ctx.slope, ctx.dims = n, float(slope), (B, Co, N, K)
        return pooled

    @staticmethod
    def backward(ctx, gp):
        lib = L.load()
        idx, y, pooled, arg, yraw, gamma, xyz0 = ctx.saved_tensors
        n, slope = ctx.norm, ctx.slope
        B, Co, N, K = ctx.dims
        dev = y.device
        gp = gp.contiguous()
        R = B * N * K
        # the two sums of the BatchNorm backward: the routed gradient is non-zero on ONE edge per (channel, point)
        nparts = max(1, min(B, 2048 // max(Co, 1)))
        part = _f32(nparts, 2, _c32(Co), device=dev)
        L.check(lib.pcr_bn_sums_f32(L.ptr(yraw), L.ptr(gp), L.ptr(n["scale"]), L.ptr(n["shift"]), 1, ctypes.c_float(slope),
                                    L.ptr(n["mean"]), 0, L.ptr(part), nparts, B, Co, N, L.stream_ptr()), "pcr_bn_sums_f32")
        k = bn_bwd_finalize(part, nparts, Co, R, gamma, n["mean"], n["invstd"], centre=n["mean"])
        g = _f32(B, Co, N * K, device=dev)
        L.check(lib.pcr_edge_pool_route_f32(L.ptr(gp), L.ptr(pooled), L.ptr(arg), ctypes.c_float(slope), L.ptr(g), B, Co, N, K,
                                            L.stream_ptr()), "pcr_edge_pool_route_f32")
        dtab = _f32(B, 2 * Co, N, device=dev)
        dwa_p = _f32(B, Co, 4, device=dev)
        L.check(lib.pcr_sa_l1_bwd_f32(L.ptr(xyz0), L.ptr(idx), L.ptr(g), L.ptr(y), L.ptr(k["ka"]), L.ptr(k["kb"]), L.ptr(k["kc"]),
                                      L.ptr(dtab), L.ptr(dwa_p), B, N, N, K, Co, L.stream_ptr()), "pcr_sa_l1_bwd_f32")
        return dtab, None, k["dgamma"], k["dbeta"], None, None



class Bmm(Function):
    """x (B,k,N), T (B,k,k) -> y[b] = T[b]^T x[b]  (torch.bmm(x^T, T)^T: the PointNet input / feature transforms)"""

    @staticmethod
    def forward(ctx, x, T):
        x, T = _dev(x), _dev(T)
        B, k, N = x.shape
        y = _f32(B, k, N, device=x.device)
        L.check(L.load().pcr_bmm_apply_f32(L.ptr(x), L.ptr(T), L.ptr(y), B, k, N, 0, L.stream_ptr()), "pcr_bmm_apply_f32")
        ctx.save_for_backward(x, T)
        return y

    @staticmethod
    def backward(ctx, g):
        x, T = ctx.saved_tensors
        B, k, N = x.shape
        g = g.contiguous()
        lib = L.load()
        dx = dT = None
        if ctx.needs_input_grad[0]:
            dx = _f32(B, k, N, device=x.device)
            L.check(lib.pcr_bmm_apply_f32(L.ptr(g), L.ptr(T), L.ptr(dx), B, k, N, 1, L.stream_ptr()), "pcr_bmm_apply_f32")
        if ctx.needs_input_grad[1]:
            dT = _f32(B, k, k, device=x.device)
            L.check(lib.pcr_bmm_dt_f32(L.ptr(x), L.ptr(g), L.ptr(dT), B, k, N, L.stream_ptr()), "pcr_bmm_dt_f32")
        return dx, dT
