"""Multi-GPU layer of the hot path: one process per GPU, independent shards of the pair batch.

The siamese forward has no data-path exchange (every pair is independent; eval-mode BatchNorm uses
running statistics; GroupNorm/LayerNorm are per sample -- SURVEY.md 8e), so ranks never talk
while computing.  torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests) is used only for: the rendezvous, the barrier that brackets a timed region, the
max-reduce of the elapsed time, and the optional gather of the per-pair logits on rank 0 (the
counterpart of mmdet's collect_results in the reference's multi_gpu_test).
"""
import os
import time

import torch
import torch.distributed as dist


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process => (0, 0, 1))"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend=None, device=None):
    """initialise the default process group when launched by torchrun with WORLD_SIZE > 1"""
    rank, local, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, **kw)
    return rank, local, world


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_range(n_items, rank, world):
    """contiguous, balanced [lo, hi) of n_items for `rank` (first n_items % world ranks get one more)"""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_pairs(s1, s2, rank=None, world=None):
    """this rank's slice of a batch of pairs held identically on every rank"""
    if rank is None:
        rank, _, world = env_world()
    lo, hi = shard_range(s1.shape[0], rank, world)
    return s1[lo:hi], s2[lo:hi]


def gather_logits(local_logits, n_total):
    """per-rank logits (shard order) -> full (n_total,) tensor on every rank, original pair order"""
    if not is_dist():
        return local_logits
    world = dist.get_world_size()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(width, dtype=local_logits.dtype, device=local_logits.device)
    pad[:local_logits.numel()] = local_logits
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return torch.cat([o[:hi - lo] for o, (lo, hi) in zip(out, sizes)])


def barrier(sync=None):
    if sync is not None:
        sync()
    if is_dist():
        dist.barrier()
    if sync is not None:
        sync()


def timed(fn, steps, warmup, sync=None, device=None):
    """`warmup` untimed + exactly `steps` timed calls of fn(), bracketed by barrier + device sync on
    both sides; returns the MAX elapsed seconds over ranks and fn's last result"""
    out = None
    for _ in range(warmup):
        out = fn()
    barrier(sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    barrier(sync)
    dt = time.perf_counter() - t0
    global LAST_RANK_SECONDS
    LAST_RANK_SECONDS = [dt]
    if is_dist():
        t = torch.tensor([dt], dtype=torch.float64, device=device or "cpu")
        every = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(every, t)                        # (kept for the report: stragglers show as a spread)
        LAST_RANK_SECONDS = [float(x.item()) for x in every]
        dt = max(LAST_RANK_SECONDS)
    return dt, out


LAST_RANK_SECONDS = []      # elapsed seconds of every rank in the last timed() region (rank order)


def broadcast_buffers(module, src=0):
    """broadcast BatchNorm running statistics from rank `src` before evaluation, as the reference's
    eval hook does (mmdet3d/core/hooks/eval_hook.py:102-108)"""
    if not is_dist():
        return
    for name, buf in module.named_buffers():
        if name.endswith("running_mean") or name.endswith("running_var"):
            dist.broadcast(buf, src)
