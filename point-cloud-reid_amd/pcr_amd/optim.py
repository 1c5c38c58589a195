"""AdamW + gradient-norm clipping as two HIP launches over every parameter tensor (include/pcr.h pcr_grad_sumsq_f32 /
pcr_adamw_step_f32; csrc/optim_kernels.hip).

What it replaces on the training path: mmcv's OptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)) followed by
torch.optim.AdamW.step() (reference configs_reid/_base_/schedules/cyclic_200e_lr3e-4.py:7-9).  Through torch that is
~10 multi-tensor launches per arithmetic step plus a host read of the norm (1.3 ms of a 21 ms iteration at the
pt128_train shape); here the norm never leaves the device.  `state_dict()` has torch.optim.AdamW's layout (per
parameter: step, exp_avg, exp_avg_sq), so checkpoints move between the two.
"""
import ctypes

import numpy as np
import torch

from . import _lib as L

_TAB = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8"), ("step_size", "<f4"),
                 ("bc2_sqrt", "<f4"), ("decay", "<f4"), ("one_m_beta1", "<f4"), ("beta2", "<f4"),
                 ("one_m_beta2", "<f4"), ("eps", "<f4"), ("pad_", "<f4")])
assert _TAB.itemsize == 72
_RING = 8


class FusedAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay, bias correction, no amsgrad) for fp32 parameters on the
    GPU; `step(max_norm=...)` also clips by the global 2-norm first, like clip_grad_norm_ + step().  Returns the
    gradient norm as a device scalar when max_norm is given (no host synchronisation)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._plan = None
        self._slot = 0

    # ------------------------------------------------------------------ layout --
    def _tensors(self):
        return [(g, p) for g in self.param_groups for p in g["params"] if p.requires_grad]

    def _build(self):
        # a plan is being replaced (add_param_group, a parameter object swapped): the step counts live in the old plan
        # only -- write them back first, or the new plan would restart bias correction against old moments
        self._sync_steps()
        ts = self._tensors()
        if not ts:
            raise L.PcrError("FusedAdamW: no parameters")
        dev = ts[0][1].device
        for _, p in ts:
            if p.device != dev or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise L.PcrError("FusedAdamW: every parameter must be a contiguous fp32 tensor on one GPU")
        chunk = L.load().pcr_opt_chunk()
        ct, cf = [], []
        for i, (_, p) in enumerate(ts):
            for first in range(0, p.numel(), chunk):
                ct.append(i)
                cf.append(first)
        tab = np.zeros(len(ts), dtype=_TAB)
        tab["p"] = [p.data_ptr() for _, p in ts]
        tab["n"] = [p.numel() for _, p in ts]
        plan = dict(ts=ts, dev=dev, tab=tab, n_chunks=len(ct),
                    chunk_tensor=torch.tensor(ct, dtype=torch.int32, device=dev),
                    chunk_first=torch.tensor(cf, dtype=torch.int32, device=dev),
                    part=torch.empty(max(len(ct), 1), dtype=torch.float64, device=dev),
                    tab_dev=torch.empty(len(ts) * _TAB.itemsize, dtype=torch.uint8, device=dev),
                    ring=[torch.empty(len(ts) * _TAB.itemsize, dtype=torch.uint8).pin_memory() for _ in range(_RING)],
                    events=[None] * _RING,
                    steps=np.zeros(len(ts), dtype=np.int64), have_state=[False] * len(ts))
        for i, (_, p) in enumerate(ts):       # state that a checkpoint brought in
            st = self.state.get(p)
            if st:
                self._adopt(plan, i, p, st)
        self._plan = plan
        return plan

    @staticmethod
    def _adopt(plan, i, p, st):
        for k in ("exp_avg", "exp_avg_sq"):
            if st[k].device != p.device or st[k].dtype != torch.float32 or not st[k].is_contiguous():
                st[k] = st[k].to(device=p.device, dtype=torch.float32).contiguous()
        plan["tab"]["m"][i], plan["tab"]["v"][i] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
        plan["steps"][i] = int(float(st["step"]))
        plan["have_state"][i] = True

    # -------------------------------------------------------------------- step --
    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        if closure is not None:
            raise L.PcrError("FusedAdamW: closures are not supported")
        plan = self._plan
        if plan is None or len(plan["ts"]) != len(self._tensors()) or \
                any(a[1] is not b[1] for a, b in zip(plan["ts"], self._tensors())):
            plan = self._build()
        ts, tab, steps = plan["ts"], plan["tab"], plan["steps"]
        live = np.zeros(len(ts), dtype=bool)
        gp = np.zeros(len(ts), dtype=np.uint64)
        for i, (_, p) in enumerate(ts):
            g = p.grad
            if g is None:
                continue
            if g.dtype != torch.float32 or g.device != p.device or g.is_sparse:
                raise L.PcrError("FusedAdamW: gradients must be dense fp32 tensors on the parameter's GPU")
            if not g.is_contiguous():
                g = p.grad = g.contiguous()
            if not plan["have_state"][i]:
                st = self.state[p]
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                self._adopt(plan, i, p, st)
            live[i] = True
            gp[i] = g.data_ptr()
        steps[live] += 1
        # per-tensor constants in double on the host, as torch's single-tensor path computes them
        lr = np.array([g["lr"] for g, _ in ts], dtype=np.float64)
        b1 = np.array([g["betas"][0] for g, _ in ts], dtype=np.float64)
        b2 = np.array([g["betas"][1] for g, _ in ts], dtype=np.float64)
        st_f = np.maximum(steps, 1).astype(np.float64)
        tab["g"] = gp
        tab["step_size"] = lr / (1.0 - b1 ** st_f)
        tab["bc2_sqrt"] = np.sqrt(1.0 - b2 ** st_f)
        tab["decay"] = 1.0 - lr * np.array([g["weight_decay"] for g, _ in ts], dtype=np.float64)
        tab["one_m_beta1"], tab["beta2"], tab["one_m_beta2"] = 1.0 - b1, b2, 1.0 - b2
        tab["eps"] = [g["eps"] for g, _ in ts]
        # the table travels through a ring of pinned buffers: the copy is asynchronous, and a buffer is rewritten
        # only after the copy that read it has completed
        k = self._slot
        self._slot = (k + 1) % _RING
        if plan["events"][k] is not None:
            plan["events"][k].synchronize()
        host = plan["ring"][k]
        host.numpy()[:] = tab.view(np.uint8)
        plan["tab_dev"].copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        plan["events"][k] = ev
        lib, stream = L.load(), L.stream_ptr()
        tab_p = ctypes.c_void_p(plan["tab_dev"].data_ptr())
        norm = None
        part = None
        if max_norm is not None:
            norm = torch.empty(1, dtype=torch.float32, device=plan["dev"])
            part = ctypes.c_void_p(plan["part"].data_ptr())
            L.check(lib.pcr_grad_sumsq_f32(tab_p, L.ptr(plan["chunk_tensor"]), L.ptr(plan["chunk_first"]),
                                           plan["n_chunks"], part, stream), "pcr_grad_sumsq_f32")
        L.check(lib.pcr_adamw_step_f32(tab_p, L.ptr(plan["chunk_tensor"]), L.ptr(plan["chunk_first"]),
                                       plan["n_chunks"], part, ctypes.c_float(max_norm if max_norm is not None else 0.0),
                                       L.ptr(norm) if norm is not None else None, stream), "pcr_adamw_step_f32")
        # the kernels write through raw pointers: tell autograd / every cache keyed by Tensor._version (inference launch
        # plans, padded biases) that these tensors changed, as an in-place torch op would
        torch.autograd.graph.increment_version([p for i, (_, p) in enumerate(ts) if live[i]])
        if part is not None and max_norm is not None and max_norm > 0:
            torch.autograd.graph.increment_version([p.grad for i, (_, p) in enumerate(ts) if live[i]])
        return norm[0] if norm is not None else None

    # -------------------------------------------------------------- checkpoints --
    def _sync_steps(self):
        if self._plan is None:
            return
        for i, (_, p) in enumerate(self._plan["ts"]):
            if self._plan["have_state"][i] and p in self.state:
                self.state[p]["step"] = torch.tensor(float(self._plan["steps"][i]), dtype=torch.float32)

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        for g in state_dict.get("param_groups", []):
            if g.get("amsgrad", False) or g.get("maximize", False):
                raise L.PcrError("FusedAdamW: a state_dict with amsgrad / maximize set cannot be honoured "
                                 "(pcr_adamw_step_f32 implements plain AdamW)")
        self._plan = None          # (the old plan's step counts must not be written over the loaded ones)
        super().load_state_dict(state_dict)
