"""mmcv-free training loop pieces for the ReID path (SURVEY.md 8e "Training", 8f row 2).

What the reference gets from mmcv / mmdet and this module restates (formulas as documented by mmcv 1.x; the
reference pins no version and has no tests at this boundary -- parity "unpinned", values fixed by this build's own
fixtures in tests/test_train_loop.py):

* optimizer  AdamW(lr, weight_decay) over every parameter (configs_reid/_base_/schedules/cyclic_*.py:7);
* OptimizerHook: grad-norm clipping (max_norm 35 or 1), optional GradientCumulativeOptimizerHook (loss / k,
  step every k iterations) (cyclic_*_accum*.py:9-11);
* CyclicLrUpdaterHook / CyclicMomentumUpdaterHook, by iteration, cosine annealing, two phases per cycle:
  [0, up) from 1 to target_ratio[0], [up, end) from target_ratio[0] to target_ratio[1]; the momentum hook drives
  AdamW's beta1 (cyclic_*.py:10-21);
* MMDistributedDataParallel(broadcast_buffers=False): one gradient exchange per step.  Here that is ONE flat
  bucket holding every gradient that exists (the reference's 24 never-used FP-module tensors have none and are
  left out), summed with a single all-reduce (RCCL over xGMI on the GPU box: 2.3 MB, direct rather than ring:
  SURVEY.md 5.8) and divided by the world size; BatchNorm statistics stay per rank, and are broadcast from rank 0
  before evaluation (shard.broadcast_buffers, eval_hook.py:102-108);
* checkpoints in mmcv's layout: {'meta': {...,'epoch','iter'}, 'state_dict': ..., 'optimizer': ...}.

The model's forward/backward is whatever `model.train_step(data, optimizer)` builds (ReIDNet: every node a HIP
launch strung together by autograd Functions -- pcr_amd/train_graph.py); for GPU parameters the update (norm, clip,
AdamW) is pcr_amd/optim.py's two launches, for host tensors (the CPU tests of this module) torch.optim.AdamW.
"""
import math

import torch
import torch.distributed as dist

from . import shard


def annealing_cos(start, end, factor):
    """mmcv.runner.hooks.lr_updater.annealing_cos: start -> end as factor goes 0 -> 1"""
    return end + 0.5 * (start - end) * (math.cos(math.pi * factor) + 1.0)


def cyclic_value(base, it, max_iters, target_ratio=(10.0, 1e-4), cyclic_times=1, step_ratio_up=0.4):
    """value at iteration `it` of mmcv's by-iteration cyclic policy (lr or momentum) around `base`"""
    per_phase = max_iters // cyclic_times
    up = int(step_ratio_up * per_phase)
    phases = ((0, up, 1.0, target_ratio[0]), (up, per_phase, target_ratio[0], target_ratio[1]))
    cur = it % per_phase
    for start, end, r0, r1 in phases:
        if start <= cur < end:
            return annealing_cos(base * r0, base * r1, (cur - start) / float(end - start))
    return base * target_ratio[1]


class GradBucket:
    """every existing gradient of `params` in one flat fp32 buffer: pack -> one all-reduce (sum) -> / world -> unpack.
    `extra`: small fp32 device vectors that must be averaged over the ranks in the same iteration (the logged loss
    scalars of mmdet's `_parse_losses`) ride in the TAIL of the same buffer, so that the iteration still has ONE
    collective -- and none of it inside the captured forward + backward (Trainer graph mode)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.live = None
        self.flat = None
        self.tail = 0

    def _layout(self, tail=None):
        tail = self.tail if tail is None else tail
        live = [p for p in self.params if p.grad is not None]
        if (self.live is None or len(live) != len(self.live) or any(a is not b for a, b in zip(live, self.live))
                or tail != self.tail):
            self.live, self.tail = live, tail
            n = sum(p.numel() for p in live)
            dev = live[0].device if live else torch.device("cpu")
            self.flat = torch.zeros(n + tail, dtype=torch.float32, device=dev)
        return self.live

    def nbytes(self):
        return 0 if self.flat is None else self.flat.numel() * 4

    def all_reduce_mean(self, extra=()):
        """-> the averaged `extra` vectors (fresh tensors; the inputs themselves on one rank)"""
        extra = [e.detach().reshape(-1).float() for e in extra]
        if not shard.is_dist():
            self._layout()
            return extra
        live = self._layout(sum(e.numel() for e in extra))
        if not live and not extra:
            return extra
        sizes = [p.numel() for p in live] + [e.numel() for e in extra]
        parts = self.flat.split(sizes) if sizes else []
        views = [v.view_as(p.grad) for v, p in zip(parts, live)]
        grads = [p.grad for p in live]
        tails = list(parts[len(live):])
        if views or tails:
            torch._foreach_copy_(views + tails, grads + extra)     # pack: one multi-tensor launch, not one copy per tensor
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())
        if views:
            torch._foreach_copy_(grads, views)
        return [t.clone() for t in tails]


class Trainer:
    """AdamW + cyclic lr / beta1 + gradient clipping + accumulation + one-bucket data-parallel exchange"""

    def __init__(self, model, max_iters, lr=3e-4, weight_decay=0.01, betas=(0.9, 0.999), grad_clip=35.0,
                 cumulative_iters=1, lr_target_ratio=(10.0, 1e-4), momentum_target_ratio=(0.85 / 0.95, 1.0),
                 cyclic_times=1, step_ratio_up=0.4, fused=None, graph=False):
        self.model = model
        self.max_iters = int(max_iters)
        self.base_lr, self.base_beta1 = float(lr), float(betas[0])
        self.grad_clip, self.cumulative_iters = grad_clip, int(cumulative_iters)
        self.lr_ratio, self.mom_ratio = tuple(lr_target_ratio), tuple(momentum_target_ratio)
        self.cyclic_times, self.step_ratio_up = cyclic_times, step_ratio_up
        params = [p for p in model.parameters()]
        if fused is None:
            fused = bool(params) and all(p.is_cuda for p in params)
        if fused:
            # on the GPU: norm + clip + AdamW are two HIP launches over every tensor (pcr_amd/optim.py)
            from .optim import FusedAdamW
            self.optimizer = FusedAdamW(params, lr=lr, weight_decay=weight_decay, betas=betas)
        else:
            # host tensors (the gloo tests of the exchange / schedule logic): torch's optimizer
            self.optimizer = torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay, betas=betas)
        self.fused = not isinstance(self.optimizer, torch.optim.AdamW)
        self.bucket = GradBucket(list(model.parameters()))
        self.iter = 0
        self.epoch = 0
        # graph=True: forward + backward of an iteration are captured ONCE into a HIP graph (after `graph_warmup` eager
        # iterations) and replayed: ~600 launches per iteration leave the host as one.  The step is bound by the launches'
        # host cost below ~128 pairs per GPU (9 ms per iteration whatever the batch; 16 pairs: 8.9 -> 4.0 ms replayed).
        # The gradient exchange and the update stay eager (their constants change every iteration).  Needs the fused
        # optimizer, cumulative_iters == 1 and batches of one fixed shape; anything else, or a failed capture, runs eager.
        self.graph = bool(graph) and self.fused and self.cumulative_iters == 1
        self.graph_warmup = 3
        self._g = None
        self._stream = None

    def current_lr(self):
        return cyclic_value(self.base_lr, self.iter, self.max_iters, self.lr_ratio, self.cyclic_times, self.step_ratio_up)

    def current_beta1(self):
        return cyclic_value(self.base_beta1, self.iter, self.max_iters, self.mom_ratio, self.cyclic_times,
                            self.step_ratio_up)

    def _set_hyper(self):
        lr, b1 = self.current_lr(), self.current_beta1()
        for g in self.optimizer.param_groups:
            g["lr"] = lr
            g["betas"] = (b1, g["betas"][1])
        return lr, b1

    def step(self, data):
        """one iteration (see _step); in graph mode on the trainer's OWN stream from the first iteration on: autograd's
        gradient-accumulation nodes remember the stream they were created on, and a node born on the default stream during an
        eager warm-up iteration (kept alive by any `out` the caller still holds) drags the default stream into the capture and
        breaks it"""
        if not self.graph:
            return self._step(data)
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            out = self._step(data)
        cur.wait_stream(self._stream)
        return out

    def _step(self, data):
        """one iteration: forward + backward of model.train_step(data, optimizer); every `cumulative_iters`
        iterations exchange gradients, clip, and apply AdamW.  Returns the outputs dict (+ lr, beta1 as floats;
        grad_norm as a 0-d tensor on the parameters' device in BOTH optimizer paths -- `float()` it at log time, which is
        the only place it costs a host synchronisation; log_vars is a LazyScalars dict with the same property)."""
        lr, b1 = self._set_hyper()
        # N > 1 ranks: mmdet's `_parse_losses` all-reduces the logged loss scalars inside train_step -- a collective in the
        # middle of the captured region.  Here the model DEFERS them (lazylog.DEFER_REDUCE) and they travel in the tail
        # of the gradient bucket: still ONE collective per iteration, none of it captured, and graph replay works on any
        # number of ranks (round 6; until round 5 N > 1 ranks fell back to the eager step).  With gradient accumulation
        # the exchange does not happen every iteration, so the scalars keep their own small all-reduce there.
        from . import lazylog
        defer = shard.is_dist() and self.cumulative_iters == 1
        # (graph_warmup >= 1 is enforced: the fused chains register their weights' bf16 images on the first EAGER
        # iteration; a capture of iteration 0 would bake their on-the-spot pack launches into every replay -- ADVICE r5)
        if self.graph and self.iter >= max(1, self.graph_warmup):
            with lazylog.defer_reduce(defer):
                got = self._graph_step(data)
            if got is not None:
                out, (plain, entries) = got
                out["lr"], out["beta1"] = lr, b1
                idx = [i for i, e in enumerate(entries) if e[3]]
                reduced = dict(zip(idx, self.bucket.all_reduce_mean([entries[i][2] for i in idx])))
                if entries:
                    out["log_vars"] = lazylog.LazyScalars.from_static(
                        plain, [(n, ints, reduced.get(i, v)) for i, (n, ints, v, _) in enumerate(entries)])
                norm = self.optimizer.step(max_norm=self.grad_clip)
                if norm is not None:
                    out["grad_norm"] = norm
                self.iter += 1
                return out
        if self.iter % self.cumulative_iters == 0:
            self.optimizer.zero_grad(set_to_none=True)
        if self.fused:
            from . import train_ops
            train_ops.prepack(self.model)      # every weight's packed images in one launch (they changed last step)
        with lazylog.defer_reduce(defer):
            out = self.model.train_step(data, self.optimizer)
        (out["loss"] / self.cumulative_iters).backward()
        out["lr"], out["beta1"] = lr, b1
        if (self.iter + 1) % self.cumulative_iters == 0:
            lv = out.get("log_vars")
            pend = lv.deferred() if isinstance(lv, lazylog.LazyScalars) else []
            reduced = self.bucket.all_reduce_mean([v for _, _, v in pend])
            if pend:
                lv.resolve(reduced)
            if self.fused:
                norm = self.optimizer.step(max_norm=self.grad_clip)     # a device scalar: no host round trip
                if norm is not None:
                    out["grad_norm"] = norm
            else:
                if self.grad_clip is not None:
                    params = [p for p in self.model.parameters() if p.grad is not None]
                    out["grad_norm"] = torch.nn.utils.clip_grad_norm_(params, self.grad_clip, norm_type=2).detach()
                self.optimizer.step()
        self.iter += 1
        return out

    # ---- HIP-graph replay of forward + backward ----
    def _graph_step(self, data):
        """-> (the iteration's outputs from the captured graph, (plain log items, static log entries)) -- the caller turns
        the entries into the iteration's log_vars after the exchange, which averages the ones flagged for it -- or None
        (graph mode switched off: the caller runs eager)"""
        from . import lazylog, train_ops
        keys = [k for k, v in data.items() if isinstance(v, (list, tuple)) and v and torch.is_tensor(v[0])]
        sig = self._graph_signature(data, keys)
        if sig is None:
            import warnings
            warnings.warn("pcr_amd.train.Trainer: this batch holds inputs a captured graph cannot take (bare tensors, "
                          "ragged lists or objects); running eager")
            self.graph, self._g = False, None
            return None
        g = self._g
        if g is not None and g["sig"] != sig:
            g = self._g = None                      # another batch shape: capture again
        try:
            if g is None:
                # static inputs: one stacked buffer per key; the model is fed lists of views into them, so that a new batch
                # enters with ONE stack launch per key and the captured graph reads fixed addresses
                static = {k: torch.stack(list(data[k])) for k in keys}
                feed = dict(data)
                for k in keys:
                    feed[k] = list(static[k].unbind(0))
                self.optimizer.zero_grad(set_to_none=True)
                train_ops.prepack(self.model, build_only=True)   # (table rebuilds -- allocations, a copy -- outside the capture)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=self._stream, capture_error_mode="relaxed"):
                    train_ops.prepack(self.model)
                    out = self.model.train_step(feed, self.optimizer)
                    out["loss"].backward()
                lv = out["log_vars"]
                ent = lv.static_entries() if isinstance(lv, lazylog.LazyScalars) else []
                plain = [(k, v) for k, v in dict.items(lv)] if not ent else \
                    [(k, v) for k, v in dict.items(lv) if v is not None]
                g = self._g = dict(sig=sig, graph=graph, static=static, keys=keys, out=out, entries=ent, plain=plain,
                                   params=[p for p in self.model.parameters()],
                                   bufs=[b for b in self.model.buffers()])
            else:
                for k in g["keys"]:
                    torch.stack(list(data[k]), out=g["static"][k])
            g["graph"].replay()
        except Exception as e:       # noqa: BLE001 -- whatever the capture could not take: this trainer runs eager from now on
            import warnings
            warnings.warn("pcr_amd.train.Trainer: HIP-graph capture failed (%s: %s); running eager" % (type(e).__name__, e))
            self.graph, self._g = False, None
            self.optimizer.zero_grad(set_to_none=True)
            return None
        # the replay wrote weights' packed images, BatchNorm statistics and gradients through captured launches: tell every
        # cache keyed by Tensor._version (the eager path's in-place ops / launches do this themselves)
        torch.autograd.graph.increment_version([b for b in g["bufs"] if b.is_floating_point()])
        # the captured outputs are STATIC tensors that the next replay overwrites: the caller gets its own copies (eager
        # iterations hand out fresh tensors too; a caller may hold several iterations' losses before reading any)
        out = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in g["out"].items()}
        return out, (g["plain"], g["entries"])

    def _graph_signature(self, data, keys):
        """everything a captured iteration has baked in and a replay cannot see change: the shape and dtype of EVERY
        element of every list input, the values of the non-tensor inputs, the arithmetic mode, the model's mode and
        which parameters take gradients.  None: the batch holds something a replay cannot refresh (a bare tensor is read
        at a captured address; a ragged list cannot be stacked)."""
        from . import engine, train_ops
        sig = []
        for k in sorted(data):
            v = data[k]
            if k in keys:
                if not all(torch.is_tensor(t) for t in v):
                    return None
                shapes = {(tuple(t.shape), t.dtype, t.device) for t in v}
                if len(shapes) != 1:
                    return None
                sig.append((k, len(v), shapes.pop()))
            elif v is None or isinstance(v, (bool, int, float, str)):
                sig.append((k, type(v).__name__, v))
            elif isinstance(v, (list, tuple)) and all(x is None or isinstance(x, (bool, int, float, str)) for x in v):
                sig.append((k, type(v).__name__, tuple(v)))
            else:
                return None
        sig.append(("precision", engine.PRECISION, "train_precision", train_ops.TRAIN_PRECISION,
                    "training", self.model.training,
                    "requires_grad", tuple(p.requires_grad for p in self.model.parameters()),
                    "stream_min_blocks", getattr(engine, "STREAM_MIN_BLOCKS", None)))
        return tuple(sig)

    # ---- checkpoints (mmcv layout) ----
    def state(self):
        return {"meta": {"epoch": self.epoch, "iter": self.iter, "pcr_amd": True},
                "state_dict": {k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                "optimizer": self.optimizer.state_dict()}

    def save(self, path):
        if shard.env_world()[0] == 0:
            torch.save(self.state(), path)
        shard.barrier()

    def load(self, path, strict=True, map_location="cpu"):
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
        sd = ckpt.get("state_dict", ckpt)
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
        self.model.load_state_dict(sd, strict=strict)
        if "optimizer" in ckpt:
            self.optimizer.load_state_dict(ckpt["optimizer"])
        meta = ckpt.get("meta", {})
        self.epoch, self.iter = int(meta.get("epoch", 0)), int(meta.get("iter", 0))
        return meta
