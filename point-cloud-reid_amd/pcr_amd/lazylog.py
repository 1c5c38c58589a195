"""log_vars without a host synchronisation per entry.

The reference's `match_forward` (mmdet3d/models/ReIDNet.py:426-435) and mmdet's `BaseDetector._parse_losses` (call
site ReIDNet.py:728) read every logged scalar with `.item()`: eight-plus device -> host round trips per training
iteration, each of which drains the launch queue.  Here the scalars of one iteration are stacked on the device,
copied to pinned host memory asynchronously, and turned into the same Python numbers (same keys, same values, ints
where the reference logs ints) the first time anybody READS the dict -- a trainer that logs every k iterations pays
one wait every k iterations, one that never looks pays none.
"""
from collections import OrderedDict

import torch

# True while a pcr_amd.train.Trainer runs the model's train_step on N > 1 ranks: the scalars mmdet's `_parse_losses`
# averages over the ranks are NOT all-reduced on the spot (`add_device(.., reduce=True)` keeps them on the device); the
# trainer averages them in the tail of its gradient bucket -- one collective per iteration, outside any graph capture.
DEFER_REDUCE = False


class defer_reduce:
    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        global DEFER_REDUCE
        self.prev, DEFER_REDUCE = DEFER_REDUCE, self.on

    def __exit__(self, *exc):
        global DEFER_REDUCE
        DEFER_REDUCE = self.prev
        return False


class LazyScalars(OrderedDict):
    """an OrderedDict whose pending entries live in device tensors until the first read"""

    def __init__(self, *a, **kw):
        self._pending = []          # (names, int_flags, host_tensor, event or None)
        super().__init__(*a, **kw)

    # ---- producer side ----
    def add_device(self, names, values, ints=None, reduce=False):
        """names: list of keys; values: 1-d tensor (same length) on any device; ints: per-key flag -> int(value);
        reduce: the values still have to be averaged over the ranks -- they stay on the device until `resolve` hands in
        the averaged vector (eager) or ride as a flagged static entry (capture)"""
        names = list(names)
        ints = list(ints) if ints is not None else [False] * len(names)
        values = values.detach()
        if reduce and not (values.is_cuda and torch.cuda.is_current_stream_capturing()):
            for n in names:
                OrderedDict.__setitem__(self, n, None)
            self._deferred = getattr(self, "_deferred", []) + [(names, ints, values)]
            return
        if values.is_cuda and torch.cuda.is_current_stream_capturing():
            # inside a HIP-graph capture (pcr_amd.train.Trainer(graph=True)): no host allocation, copy or event may be
            # recorded; the stacked values stay in their (static) device tensor and the trainer turns them into an
            # ordinary pending entry after every replay (`from_static`)
            for n in names:
                OrderedDict.__setitem__(self, n, None)
            self._static = getattr(self, "_static", []) + [(names, ints, values, bool(reduce))]
            return
        if values.is_cuda:
            host = torch.empty(values.shape, dtype=values.dtype).pin_memory()
            host.copy_(values, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host, ev = values.clone(), None
        for n in names:                     # keeps the reference's key order
            OrderedDict.__setitem__(self, n, None)
        self._pending.append((names, ints, host, ev))

    def static_entries(self):
        """(names, ints, device tensor, reduce flag) recorded during a graph capture"""
        return list(getattr(self, "_static", []))

    def deferred(self):
        """(names, ints, device tensor) of the entries that wait for their average over the ranks"""
        return list(getattr(self, "_deferred", []))

    def resolve(self, reduced):
        """`reduced`: one averaged vector per deferred() entry, in order"""
        pend, self._deferred = self.deferred(), []
        for (names, ints, _), v in zip(pend, reduced):
            self.add_device(names, v, ints)

    @classmethod
    def from_static(cls, plain, entries):
        """a fresh LazyScalars after a graph replay: `plain` = the already-known items, `entries` = static_entries() of
        the captured dict, whose device tensors now hold this replay's values (copied out asynchronously here)"""
        out = cls()
        for k, v in plain:
            OrderedDict.__setitem__(out, k, v)
        for e in entries:
            out.add_device(e[0], e[2], e[1])
        return out

    # ---- consumer side ----
    def materialize(self):
        if getattr(self, "_deferred", None):
            # nobody averaged them (a train_step outside a Trainer while DEFER_REDUCE was set): this rank's own values
            self.resolve([v for _, _, v in self.deferred()])
        pend, self._pending = self._pending, []
        for names, ints, host, ev in pend:
            if ev is not None:
                ev.synchronize()
            vals = host.tolist()
            for n, i, v in zip(names, ints, vals):
                if OrderedDict.__contains__(self, n) and OrderedDict.__getitem__(self, n) is None:
                    OrderedDict.__setitem__(self, n, int(round(v)) if i else v)
        return self

    def __getitem__(self, k):
        self.materialize()
        return OrderedDict.__getitem__(self, k)

    def get(self, k, default=None):
        self.materialize()
        return OrderedDict.get(self, k, default)

    def items(self):
        self.materialize()
        return OrderedDict.items(self)

    def values(self):
        self.materialize()
        return OrderedDict.values(self)

    def copy(self):
        self.materialize()
        return OrderedDict(OrderedDict.items(self))

    def update(self, other=(), **kw):
        if isinstance(other, LazyScalars):          # keep the other's entries lazy too
            if getattr(other, "_static", None):
                self._static = getattr(self, "_static", []) + other._static
            if getattr(other, "_deferred", None):
                self._deferred = getattr(self, "_deferred", []) + other._deferred
                other._deferred = []
            pend, other._pending = other._pending, []
            for n in OrderedDict.keys(other):
                OrderedDict.__setitem__(self, n, OrderedDict.__getitem__(other, n))
            self._pending.extend(pend)
            other = ()
        OrderedDict.update(self, other, **kw)

    def __repr__(self):
        self.materialize()
        return OrderedDict.__repr__(self)

    def __reduce__(self):
        self.materialize()
        return (OrderedDict, (list(OrderedDict.items(self)),))
