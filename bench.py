"""Throughput bench of the siamese point-cloud ReID hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--pairs B]

One step = one pass of the hot path (siamese_forward + match_forward_inference)
over one batch of B synthetic pairs that is already resident in HBM.  Every rank (one process
per GPU, RCCL only for the barrier / max-reduce of the timing) runs its own independent shard of
pairs -- the path has no data-path collective -- so scaling is weak and `value` is the whole-job
pairs/s.  Prints ONE JSON line on rank 0 (see the task contract) carrying `roofline` for the
dominant kernel and, at N=1, `cpu_baseline` (the torch restatement oracle/model_oracle.py, which
is pinned to the reference by tests/golden, timed on this box's host cores on a bounded sample).
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "point-cloud-reid_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

PT_MODEL = dict(
    type="ReIDNet", hidden_size=128, combine="point-cat", match_type="xcorr_eff", pool_type="both",
    backbone_list=[128, 64, 32], output_sequence_size=64,
    backbone=dict(type="Pointnet_Backbone", input_channels=0, use_xyz=True, conv_out=64),
    match_head=[dict(type="LinearRes", n_in=128, n_out=128, norm="GN", ng=8),
                dict(type="Linear", in_features=128, out_features=1)],
    downsample=None, cls_head=None, fp_head=None, shape_head=None,
    cross_stage1=dict(type="corss_attention", d_model=64, nhead=2, attention="linear"),
    cross_stage2=dict(type="corss_attention", d_model=64, nhead=2, attention="linear"),
    local_stage1=dict(), local_stage2=dict(),
    losses_to_use=dict(kl=False, match=True, cls=False, shape=False, fp=False, triplet=False))

# name -> (description, model kind, points, backbone_list, default pairs per GPU per step)
WORKLOADS = {
    "pt1024": ("Point-Transformer ReIDNet (configs_reid/reid_nuscenes_pts/num_point_ablation_test/"
               "pts_point-transformer_r_nus_det_400e_1024pts.py), 1024-pt synthetic pairs, eval", "pt", 1024,
               [1024, 512, 256], 512),
    "pt128": ("Point-Transformer ReIDNet (reid_nuscenes_pts/testing_pts_point-transformer_r_nus_det_500e.py), "
              "128-pt synthetic pairs, eval", "pt", 128, [128, 64, 32], 512),
    "ssg1024": ("PointNet++ SSG siamese (BASELINE config 2; SA(512,r.2,K32,[64,64,128]) -> SA(128,r.4,K64,[128,128,256]) "
                "-> Conv1d 64; D-FPS + ball query), 1024-pt synthetic pairs, eval", "ssg", 1024, None, 4096),   # (2048 until
    # the end of round 4: 4096 pairs per pass measure 4-5 % more pairs/s -- launch tails amortised -- 8192 fewer again)
    "pointnet256": ("PointNet ReIDNet (configs_reid/_base_/reidentifiers/reid_pts_pointnet_point-cat.py), 256-pt "
                    "synthetic pairs, eval (BASELINE config 1 shape)", "pointnet", 256, None, 1024),   # (256 pairs per pass until the end
    # of round 4: 62.9 k pairs/s at 256, 67.0 k at 512, 71.2 k at 1024 on one box)
    "gallery128": ("amortised gallery matching (SURVEY 8f rank 1; forward_inference ReIDNet.py:189-191 + "
                   "match_forward_inference :444-462, the tracker use-case): G tracks x G detections of 128 pts -- every "
                   "object is encoded ONCE, then all G*G combinations go through the matching head (match_gallery); "
                   "`pairs` = G*G comparisons per GPU per step", "gallery", 128, [128, 64, 32], 192 * 192),
    "pt128_train": ("Point-Transformer siamese TRAINING step (BASELINE config 4 shape: nuScenes-ReID 128-pt crops, 256 pairs "
                    "per GPU): forward + backward + one-bucket gradient all-reduce + clip + AdamW (cyclic lr/beta1). Forward "
                    "(BatchNorm batch statistics), backward and the update (norm + clip + AdamW) are HIP launches end to end "
                    "(pcr_amd/train_ops.py, optim.py)", "pt_train", 128,
                    [128, 64, 32], 256),
    "ptxcorr128": ("Point-Transformer with the baseline-orig matching (match_type='xcorr': cross -> local_self_attention "
                   "-> cross -> local; reid_waymo_pts/testing_pts_point-transformer_baseline-orig_r_waymo_det_400e.py), "
                   "128-pt synthetic pairs, eval", "ptx", 128, [128, 64, 32], 512),
    "dgcnn128": ("DGCNN ReIDNet (reid_waymo_pts/testing_pts_dgcnn_r_waymo_det_400e.py: 128-pt crops, 512 pairs per "
                 "GPU), synthetic pairs, eval", "dgcnn", 128, None, 512),
    "dgcnn256": ("DGCNN ReIDNet (configs_reid/_base_/reidentifiers/reid_pts_dgcnn_point-cat.py, k=20, emb 1024), "
                 "256-pt synthetic pairs, eval", "dgcnn", 256, None, 512),
    "dgcnn1024": ("DGCNN ReIDNet (reid_waymo_pts/num_point_ablation_test/pts_dgcnn_r_waymo_det_400e_1024pts.py), "
                  "1024-pt synthetic pairs, eval", "dgcnn", 1024, None, 128),
    "pt4096": ("Point-Transformer ReIDNet, 4096-pt Waymo-shape synthetic pairs, eval", "pt", 4096,
               [4096, 2048, 1024], 256),
}

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3    # f32-input MFMA dense peak
MFMA_BF16_PEAK_TF = 2500.0  # bf16 MFMA dense peak (MI355X_MICROARCH.md: ~2.5 PF dense; 16x the f32-input rate)
# arithmetic of the grouped-SA matrix layers (pcr_amd/engine.py PRECISION): (dtype label, peak, MFMAs issued per product)
PREC_INFO = {"f32": ("f32", MFMA_F32_PEAK_TF, 1), "bf16x3": ("bf16x3", MFMA_BF16_PEAK_TF, 3),
             "bf16": ("bf16", MFMA_BF16_PEAK_TF, 1)}


def build_pt_model(backbone_list, device="cuda"):
    from mmdet3d.models import build_model
    from pcr_amd import testing as T
    cfg = copy.deepcopy(PT_MODEL)
    cfg["backbone_list"] = list(backbone_list)
    model = build_model(cfg)
    man = T.load_manifest(os.path.join(ROOT, "tests", "golden", "pt_manifest.json"))
    sd = T.seeded_state_dict(man, 0)
    model.load_state_dict(sd, strict=True)
    return model.to(device).eval(), sd


SSG_MODEL = copy.deepcopy(PT_MODEL)
SSG_MODEL["backbone"] = dict(type="PointNet2SSG", num_points=(512, 128), radii=(0.2, 0.4), num_samples=(32, 64),
                             sa_channels=((64, 64, 128), (128, 128, 256)), conv_out=64)


PN_MODEL = copy.deepcopy(PT_MODEL)
PN_MODEL.update(use_dgcnn=True, backbone=dict(type="PointNet", k=40, normal_channel=False),
                downsample=[dict(type="LinearRes", n_in=1024, n_out=512, norm="GN", ng=64),
                            dict(type="LinearRes", n_in=512, n_out=128, norm="GN", ng=16),
                            dict(type="Linear", in_features=128, out_features=64)])


DG_MODEL = copy.deepcopy(PN_MODEL)
DG_MODEL.update(backbone=dict(type="dgcnn", dropout=0.5, emb_dims=1024, k=20, output_channels=40))
DG_MODEL["match_head"] = [dict(type="LinearRes", n_in=128, n_out=128, norm="GN", ng=16),
                          dict(type="Linear", in_features=128, out_features=1)]


LOCAL_STAGE = dict(type="local_self_attention", d_model=64, nhead=2, attention="linear", knum=48, pos_size=64)


def build_model(kind, backbone_list, device="cuda"):
    """kind 'pt' (reference Point-Transformer config), 'ptx' (the same with the baseline-orig matching), 'pointnet',
    'dgcnn' (reference configs) or 'ssg' (BASELINE config 2 composition); seeded weights"""
    if kind == "pt":
        return build_pt_model(backbone_list, device)
    if kind == "ptx":
        from mmdet3d.models import build_model as _build
        from pcr_amd import testing as T
        cfg = copy.deepcopy(PT_MODEL)
        cfg.update(match_type="xcorr", local_stage1=dict(LOCAL_STAGE), local_stage2=dict(LOCAL_STAGE),
                   backbone_list=list(backbone_list))
        model = _build(cfg)
        sd = T.seeded_state_dict(T.manifest_of(model), 0)
        model.load_state_dict(sd, strict=True)
        return model.to(device).eval(), sd
    from mmdet3d.models import build_model as _build
    from pcr_amd import testing as T
    model = _build(copy.deepcopy({"ssg": SSG_MODEL, "dgcnn": DG_MODEL}.get(kind, PN_MODEL)))
    sd = T.seeded_state_dict(T.manifest_of(model), 0)
    model.load_state_dict(sd, strict=True)
    return model.to(device).eval(), sd


def hot_path(model, s1, s2):
    xyz1, xyz2, h1, h2 = model.siamese_forward(s1, s2)
    return model.match_forward_inference(h1, h2, xyz1, xyz2)


def aggregate_profile(rec, detail=True, elapsed=None):
    """engine.PROFILE records -> {launch name: [ms, calls, reference flops, algorithmic bytes, issued flops, arithmetic]}.
    `arithmetic` is what the LIBRARY says the launch ran its matrix phases in (engine._prof, pcr_last_launch_arith), None
    for launches that do not multiply."""
    elapsed = elapsed or (lambda e0, e1: e0.elapsed_time(e1))
    tot = {}
    for name, e0, e1, flops, nbytes, exec_flops, arith in rec:
        if not detail:
            name = name.split("[")[0]
        t = tot.setdefault(name, [0.0, 0, 0.0, 0.0, 0.0, arith])
        t[0] += elapsed(e0, e1)
        t[1] += 1
        t[2] += flops
        t[3] += nbytes
        t[4] += exec_flops
        t[5] = arith
    return tot


def profile_kernels(model, s1, s2, reps=3, detail=False, fn=None):
    """per-launch device time with events on the launch stream; returns the dominant launch"""
    from pcr_amd import engine
    best = None
    for _ in range(reps):
        engine.PROFILE = []
        with torch.no_grad():
            if fn is not None:
                fn()
            else:
                hot_path(model, s1, s2)
        torch.cuda.synchronize()
        rec = engine.PROFILE
        engine.PROFILE = None
        tot = aggregate_profile(rec, detail)
        if best is None or sum(v[0] for v in tot.values()) < sum(v[0] for v in best.values()):
            best = tot
    return best


def cpu_baseline(workload, sd, budget_s=20.0):
    import model_oracle as MO
    from pcr_amd import testing as T
    _, kind, n, bl, _ = WORKLOADS[workload]
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    pairs = 8 if n >= 1024 else 32
    s1, s2 = T.synthetic_pairs(pairs, n, seed=1234, kind="box" if kind == "ssg" else "randn")
    run = {"ssg": lambda: MO.ssg_pairs(sd, s1, s2), "pointnet": lambda: MO.pointnet_pairs(sd, s1, s2),
           "pt": lambda: MO.pt_pairs(sd, s1, s2, bl), "dgcnn": lambda: MO.dgcnn_pairs(sd, s1, s2),
           "ptx": lambda: MO.pt_pairs_xcorr(sd, s1, s2, bl)}[kind]
    best, best_threads, runs = None, 1, 0
    t_start = time.time()
    # torch's intra-op pool does not scale to hundreds of threads on these small per-cloud ops:
    # try a few pool sizes up to the cores we may use and report the fastest
    candidates = sorted({min(avail, c) for c in (8, 16, 32, avail)})
    with torch.no_grad():
        for threads in candidates:
            torch.set_num_threads(threads)
            for _ in range(2):
                if time.time() - t_start > budget_s and runs >= 2:
                    break
                t0 = time.time()
                run()
                dt = time.time() - t0
                runs += 1
                if best is None or dt < best:
                    best, best_threads = dt, threads
    # `cores` (the contract's field) = the threads the reported run actually used; `host_cores` = what this box offers
    return dict(value=pairs / best, unit="pairs/s", cores=best_threads, threads_used=best_threads, host_cores=avail,
                kind="port",
                sample="%d pairs x %d pts, best of %d runs over thread counts %s (%d usable cores), torch %s eager "
                       "fp32 restatement of the reference graph (oracle/model_oracle.py)"
                       % (pairs, n, runs, candidates, avail, torch.__version__))


def train_bench(args, desc, n, bl, pairs, rank, world, steps=None, warmup=None):
    """training throughput (reported separately from the inference metric): pairs/s of
    Trainer.step = ReIDNet.train_step forward + backward, ONE flat-bucket gradient all-reduce over RCCL, gradient
    clipping and AdamW with the cyclic schedule, on a fixed synthetic batch already resident in HBM"""
    from pcr_amd import shard, train, train_ops
    from pcr_amd import testing as T
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    model, _ = build_pt_model(bl)
    model.train()
    s1, s2 = T.synthetic_pairs(pairs, n, seed=4321 + rank, kind="randn")
    dev = "cuda"
    g = torch.Generator().manual_seed(99 + rank)
    ids1 = torch.arange(pairs)
    ids2 = torch.where(torch.rand(pairs, generator=g) < 0.5, ids1, ids1 + pairs)      # half of the pairs match
    zero = torch.zeros(1, dtype=torch.long, device=dev)
    data = dict(sparse_1=list(s1.to(dev)), sparse_2=list(s2.to(dev)), dense_1=list(s1.to(dev)), dense_2=list(s2.to(dev)),
                label_1=[zero] * pairs, label_2=[zero] * pairs,
                id_1=[i.view(1).to(dev) for i in ids1], id_2=[i.view(1).to(dev) for i in ids2])
    # forward + backward replayed from a HIP graph (Trainer(graph=True)); the exchange and the update stay eager.  Until round
    # 4 the GPU was the bound at this workload's 256 pairs either way (10.4-10.9 ms replayed against 10.45 eager) and eager
    # was the default.  (Round 5: the fused chains brought the step's GPU time down to about what the host needs to issue its ~400 launches --
    # inside the default run, after nine other workloads, the eager step measured 10.4 ms against 9.2 alone.  One rank
    # therefore times the REPLAYED iteration by default, a fixed mode like the inference lines'; PCR_TRAIN_GRAPH=0 keeps
    # one launch per node.  Round 6: N > 1 ranks replay too -- the logged loss scalars that mmdet all-reduces inside
    # train_step ride in the tail of the gradient bucket, so the captured region holds no collective -- and config.launch
    # names what ran)
    tr = train.Trainer(model, max_iters=steps + warmup + 3, lr=3e-4, grad_clip=1.0,
                       graph=os.environ.get("PCR_TRAIN_GRAPH", "1") == "1" and warmup >= 2)
    tr.graph_warmup = 1          # iteration 0 eager, iteration 1 captures: both inside the W warm-up steps
    prewarm()
    # (a pilot of both ways like the inference lines' was tried: an eager step AFTER a replay runs on the trainer's own
    # stream with autograd's cross-stream synchronisation and measures 13 ms against 10.3 -- not a fair pilot; eager stays)
    dt, out = shard.timed(lambda: tr.step(data)["loss"].detach(), steps, warmup,
                          sync=torch.cuda.synchronize, device="cuda")
    assert torch.isfinite(out).all()
    per_rank = [t / steps * 1e3 for t in shard.LAST_RANK_SECONDS]
    # per-launch device times of one more step (events on the launch stream): the dominant TRAINING launch.  EVERY rank
    # runs that step (it contains the gradient all-reduce: rank 0 alone would wait for the others forever); only rank
    # 0 records it
    from pcr_amd import engine
    clk = clock_probe() if rank == 0 else None
    if rank == 0:
        engine.PROFILE = []
    graphed = tr.graph
    tr.graph = False             # (the per-launch events need the launches themselves: this one step runs eager)
    tr.step(data)
    torch.cuda.synchronize()
    line = None
    if rank == 0:
        tr.bucket._layout()
        rec, engine.PROFILE = engine.PROFILE, None
        tot = {}
        for name, e0, e1, flops, nbytes, _, arith in rec:
            t = tot.setdefault(name, [0.0, 0, 0.0, 0.0, arith])
            t[0] += e0.elapsed_time(e1)
            t[1] += 1
            t[2] += flops
            t[3] += nbytes
        dom = max(tot, key=lambda k: tot[k][0] / tot[k][1])
        ms, cnt, flops, nbytes, dom_arith = tot[dom]
        dom_arith = dom_arith or "f32"
        _, mfma_peak, mult = PREC_INFO[dom_arith]
        groups = {}
        for k, v in tot.items():
            groups[k.split("[")[0]] = groups.get(k.split("[")[0], 0.0) + v[0]
        # (split bf16 issues three MFMAs per product: all three counted against the bf16 peak)
        t_mfma, t_hbm = mult * flops / cnt / (mfma_peak * 1e12), nbytes / cnt / (HBM_PEAK_GBS * 1e9)
        if t_mfma >= t_hbm:
            roof = dict(bound="mfma", achieved=mult * (flops / cnt) / (ms / cnt * 1e-3) / 1e12, peak=mfma_peak, unit="TFLOP/s")
        else:
            roof = dict(bound="hbm", achieved=(nbytes / cnt) / (ms / cnt * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s")
        roof.update(kernel=dom, kernel_arithmetic=dom_arith, frac=roof["achieved"] / roof["peak"], avg_launch_ms=ms / cnt,
                    launches_per_step=cnt, traffic=None, algorithmic_gflop_per_launch=flops / cnt / 1e9,
                    algorithmic_mb_per_launch=nbytes / cnt / 1e6,
                    profiled_kernels_ms={k: round(v, 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])},
                    note="the step has ~450 launches; `profiled_kernels_ms` covers the train-dense / grouped-SA launches "
                         "(events on the launch stream), the rest (norms, attention core, reductions, packing, AdamW) is in "
                         "ms_per_step only")
        add_clock(roof, clk)
        line = {
            "metric": "siamese training pairs/sec @%d pts" % n, "value": world * pairs * steps / dt,
            "unit": "pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "per_rank_ms_per_step": per_rank, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32",
                      "bf16x3": "f32 forward; split bf16 (three bf16 MFMAs per product) for the gradient products of the "
                                "128 x 128 grouped-MLP layers and of the fused attention chains",
                      "bf16x3_all": "f32; split bf16 for the fused attention chains (forward too) and the gradient "
                                    "products of the 128 x 128 grouped-MLP layers"}[train_ops.TRAIN_PRECISION],
            "data": "synthetic (randn clouds, seeded random-init weights)",
            "config": {"workload": "pt128_train: %s" % desc, "pairs_per_gpu_per_step": pairs, "points": n,
                       "backbone_list": bl, "parallelism": "data parallel x%d, one %d-byte gradient bucket per step"
                       % (world, tr.bucket.nbytes()), "rccl_ranks": world,
                       "launch": "hipgraph: forward + backward replayed from one HIP graph, exchange + update eager"
                       if graphed else "eager (one launch per node)"},
            "roofline": roof}
    del model, tr, data
    torch.cuda.empty_cache()
    return line


def clock_probe(iters=20000):
    """GHz the matrix core sustains with every CU busy on f32 MFMAs (pcr_clock_probe), measured right after the
    timed region (warm chip).  Two readings: 16-MFMA rounds counted against the constant-rate wall clock (the one
    reported), and the shader-clock counter against the same wall clock (a cross-check)."""
    import ctypes
    from pcr_amd import _lib as L
    lib = L.load()
    n_cu = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    ticks = torch.zeros((n_cu, 2), dtype=torch.int64, device="cuda")
    khz = lib.pcr_wall_clock_khz()
    if khz <= 0:
        return None
    for _ in range(2):    # first call: code load
        L.check(lib.pcr_clock_probe(L.ptr(ticks), n_cu, iters, L.stream_ptr()), "pcr_clock_probe")
    torch.cuda.synchronize()
    t = ticks.cpu().double()
    wall_s = t[:, 1] / (khz * 1e3)
    by_mfma = (iters * 16 * 64) / wall_s / 1e9
    by_counter = t[:, 0] / wall_s / 1e9
    return dict(clock_ghz=float(by_mfma.median()), shader_counter_ghz=float(by_counter.median()),
                wall_clock_khz=khz)


_PREWARMED = False


LAST_PILOT = {}              # {"eager_ms": .., "hipgraph_ms": ..} of the last graph_step() (both go into the record)


def graph_step(fn):
    """fn (a forward pass over resident inputs, no host round trip inside) captured ONCE into a HIP graph and replayed:
    -> (callable returning fn's output tensor, "hipgraph") or (fn, "eager: <why>").  Same launches, same bits -- the
    replay is checked against an eager call before it is used -- but the step no longer depends on how fast this box's
    host can issue ~25 launches and their tensor allocations (a loaded host of the pool: 6.7 ms eager against 5.1).
    The mode is FIXED, never picked by a race: PCR_BENCH_GRAPH=1 (the default) times the replay, PCR_BENCH_GRAPH=0 the
    eager loop; an untimed three-step pilot of BOTH ways is recorded beside the result (LAST_PILOT -> config.eager_ms /
    config.hipgraph_ms) so that lines of either mode stay comparable."""
    LAST_PILOT.clear()
    want_graph = os.environ.get("PCR_BENCH_GRAPH", "1") != "0"

    def pilot(f, n=3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(n):
                f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    try:
        if not torch.cuda.is_available():
            raise RuntimeError("no device to capture on")
        with torch.no_grad():
            ref = None
            for _ in range(2):                      # lazy plans, LDS attributes, allocator pools: all set up eagerly
                ref = fn()
            LAST_PILOT["eager_ms"] = pilot(fn)      # (BEFORE the capture: the first eager calls after one are not the steady state)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):   # (other threads -- RCCL's watchdog -- may call HIP meanwhile)
                out = fn()
            g.replay()
            torch.cuda.synchronize()
            if out.shape != ref.shape or not torch.equal(out, ref):
                raise RuntimeError("the replay's output differs from the eager call's")

        def run():
            g.replay()
            return out
        run._graph = g
        LAST_PILOT["hipgraph_ms"] = pilot(run)
        if not want_graph:
            del run, g
            return fn, "eager: PCR_BENCH_GRAPH=0"
        return run, "hipgraph"
    except Exception as e:                          # (capture is an optimisation of the MEASUREMENT, never a requirement)
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        if not want_graph:
            return fn, "eager: PCR_BENCH_GRAPH=0"
        return fn, "eager: capture failed (%s: %s)" % (type(e).__name__, str(e)[:120])


def prewarm(seconds=0.25):
    """A freshly leased GPU runs its first ~150 ms at a fraction of its clock (measured: the first 20 steps of a cold
    process took 2.2x the time of the same steps after 50 warm-up steps, pt128).  Before the W warm-up steps of the
    contract the device therefore spins on the clock-probe kernel (all CUs, no workload data touched) for about
    `seconds`; the W warm-up steps and the K timed steps are unchanged."""
    global _PREWARMED
    if _PREWARMED:
        return
    _PREWARMED = True
    t0 = time.time()
    while time.time() - t0 < seconds:
        if clock_probe(iters=20000) is None:
            break


def ssg_fill(model, s1):
    """mean genuine ball-query hits / K of every SA layer on these clouds (how much of the reference's K-row work
    the duplicate-free SA evaluation really has to do)"""
    from mmdet3d.ops.point_ops import ball_query_cnt, furthest_point_sample, gather_points
    xyz = s1.contiguous()
    fill = {}
    with torch.no_grad():
        for i, sa in enumerate(model.backbone.SA_modules):
            idx = furthest_point_sample(xyz, sa.num_point[0])
            new_xyz = gather_points(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
            g = sa.groupers[0]
            _, cnt = ball_query_cnt(g.min_radius, g.max_radius, g.sample_num, xyz, new_xyz)
            fill["sa%d" % (i + 1)] = dict(K=g.sample_num, radius=g.max_radius,
                                          mean_hits=float(cnt.float().mean()),
                                          fill=float(cnt.float().mean()) / g.sample_num)
            xyz = new_xyz
    return fill


HBM_LAUNCHES = ("knn_prefix", "fps", "ball_query", "pool_head", "gather", "edge_max", "knn_feat", "local_attn")


def roofline_object(prof):
    """{launch: [ms, calls, reference flops, bytes, issued flops, arithmetic]} (aggregate_profile) -> the roofline object of
    the single most expensive LAUNCH.  Pure (tests/test_bench_roofline.py feeds it synthetic profiles).
    achieved = FLOPs the launch really issues on the matrix core / its duration: the kernel skips work the reference does
    (first MLP layer via per-point tables, repeated ball-query rows), and in split bf16 every product is three MFMAs, all
    three counted and priced against the bf16 peak; `product_tflops` is the same launch per PRODUCT, the reference's op
    count over the same time is `reference_op_tflops`.  The peak is the peak of the arithmetic the launch RAN
    (prof[..][5], reported by the library per launch), never of the mode that was asked for."""
    dom = max(prof, key=lambda k: prof[k][0] / prof[k][1])
    ms, cnt, flops, nbytes, exec_flops, arith = prof[dom]
    step_ms_kern = sum(v[0] for v in prof.values())
    groups = {}
    for k, v in prof.items():
        groups[k.split("[")[0]] = groups.get(k.split("[")[0], 0.0) + v[0]
    sec = ms / cnt * 1e-3
    per = {k: round(v, 4) for k, v in sorted(groups.items(), key=lambda kv: -kv[1])}
    if arith is None or dom.split("[")[0] in HBM_LAUNCHES:
        # neighbour search / sampling / pooling launches move bytes, they do not multiply: priced against HBM with
        # their ALGORITHMIC bytes (read xyz, write indices) -- their real limiter today is instruction
        # issue, which this fraction makes plain
        roof = dict(kernel=dom, bound="hbm", achieved=(nbytes / cnt) / sec / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                    avg_launch_ms=ms / cnt, launches_per_step=cnt, traffic=None, kernel_arithmetic=arith,
                    algorithmic_mb_per_launch=nbytes / cnt / 1e6, share_of_step=ms / step_ms_kern, per_kernel_ms=per)
    else:
        _, peak, mult = PREC_INFO[arith]
        roof = dict(kernel=dom, bound="mfma", achieved=(exec_flops * mult / cnt) / sec / 1e12,
                    peak=peak, unit="TFLOP/s", avg_launch_ms=ms / cnt, launches_per_step=cnt, traffic=None,
                    kernel_arithmetic=arith, mfma_per_product=mult,
                    product_tflops=(exec_flops / cnt) / sec / 1e12,
                    product_frac_of_f32_mfma_peak=(exec_flops / cnt) / sec / 1e12 / MFMA_F32_PEAK_TF,
                    issued_gflop_per_launch=exec_flops * mult / cnt / 1e9,
                    reference_op_gflop_per_launch=flops / cnt / 1e9,
                    reference_op_tflops=(flops / cnt) / sec / 1e12,
                    share_of_step=ms / step_ms_kern, per_kernel_ms=per)
    roof["frac"] = roof["achieved"] / roof["peak"]
    return roof


def attach_pmc(roof, workload, pairs, precision):
    """HBM bytes per launch and matrix-pipe occupancy of THE kernel the roofline object describes, from the committed
    rocprofv3 --pmc passes (profiles/rNN_<workload>_pmc.json, tools/pmc_summary.py: FETCH_SIZE and WRITE_SIZE in
    separate passes, gfx950 correction 2 * FETCH_SIZE + WRITE_SIZE), scaled to this batch size.  Only a profile taken
    in the SAME arithmetic mode and mapping this launch to a kernel of the SAME arithmetic is quoted; otherwise the
    fields stay null."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_pmc.json" % workload)), reverse=True):
        try:
            with open(path) as f:
                pmc = json.load(f)
            if pmc.get("_precision") != precision:      # (files older than round 4 do not say: never quoted)
                continue
            kern = pmc["_launch_to_kernel"][roof["kernel"]]
            ent = pmc[kern]
            if ent.get("arithmetic") not in (None, roof.get("kernel_arithmetic")):
                continue
            roof["traffic"] = ent["hbm_bytes_corrected"] * pairs / pmc["_pairs_per_step"]
            roof["traffic_source"] = "%s (%s)" % (os.path.relpath(path, ROOT), kern)
            roof["mfma_pipe_busy_pmc"] = ent["mfma_pipe_busy"]
            roof["pmc_launch_ms"] = ent["launch_us"] / 1e3 * pairs / pmc["_pairs_per_step"]
            return roof
        except (OSError, KeyError, TypeError, IndexError, ValueError):
            continue
    return roof


def roofline_of(model, s1, s2, workload, pairs, fn=None):
    """per-launch device times (events on the launch stream); the roofline object describes the single most
    expensive LAUNCH"""
    from pcr_amd import engine
    prof = profile_kernels(model, s1, s2, detail=True, fn=fn)
    roof = attach_pmc(roofline_object(prof), workload, pairs, engine.PRECISION)
    return roof, prof


def precision_text(mode):
    """what `dtype` means, launch family by launch family (the per-launch truth is roofline.kernel_arithmetic)"""
    if mode == "f32":
        return "f32: every matrix phase on the f32-input MFMA (exact fmaf chains, the reference's arithmetic)"
    how = {"bf16x3": "split bf16 (hi + lo operands, three bf16 MFMAs per product, f32 accumulate)",
           "bf16": "plain bf16 operands (one MFMA per product, f32 accumulate)"}[mode]
    return ("%s: grouped-SA layers 2/3 and the layer-1 tables as %s; attention (d_model <= 128) projections, message, "
            "feed-forward and cov_final as split bf16 in both bf16 modes; f32-input MFMA for the coordinate part of SA "
            "layer 1, the KV accumulation / merge fold, the tile kv kernel (d = 128 / 96), wide attention (d > 128) and "
            "the Conv1d / PointNet / DGCNN dense layers; pooling + match head in f32 VALU" % (mode, how))


def add_clock(roof, clk):
    """the clock beside frac: the spec peak assumes 2.4 GHz; under sustained MFMA load the chip runs lower"""
    if clk is None:
        return
    roof["clock_ghz"] = clk["clock_ghz"]
    roof["clock_source"] = ("pcr_clock_probe: sustained v_mfma_f32_32x32x2_f32 on all CUs right after the timed "
                            "region, MFMA count / wall clock (shader counter cross-check %.3f GHz)"
                            % clk["shader_counter_ghz"])
    if roof["bound"] == "mfma" and roof.get("kernel_arithmetic", "f32") == "f32":   # (the probe runs f32 MFMAs)
        roof["peak_at_clock"] = roof["peak"] * clk["clock_ghz"] / 2.4
        roof["frac_at_clock"] = roof["achieved"] / roof["peak_at_clock"]


def add_sustained(roof):
    """informational, beside the contract's nominal `peak` / `frac` (which stay as they are): what a dense
    v_mfma_f32_32x32x16_bf16 loop on all CUs SUSTAINS when its operands are random numbers instead of constants -- the chip
    lowers its shader clock under that load (tools/probe_clock.hip; committed output profiles/r*_clock_probe.txt: mode 0 =
    constant operands = the nominal 2.5 PFLOP/s at 2.39 GHz, mode 1 = random operands, mode 7 = random operands fetched from
    LDS with a quarter of the issue slots VALU work, the mix of the K-row SA kernel)"""
    import glob
    import re
    if roof.get("bound") != "mfma" or not str(roof.get("kernel_arithmetic", "")).startswith("bf16"):
        return
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_clock_probe.txt")), reverse=True):
        try:
            best = {}
            with open(path) as f:
                for ln in f:
                    m = re.match(r"mode (\d)\s+launch\s+([\d.]+) ms\s+([\d.]+) TFLOP/s.*median\s+(\d+)", ln)
                    if m and float(m.group(2)) > 50.0:            # the long launches
                        best[int(m.group(1))] = (float(m.group(3)), int(m.group(4)))
            if 1 in best and 0 in best:
                roof["sustained_bf16"] = {
                    "constant_operands_tflops": best[0][0], "random_operands_tflops": best[1][0],
                    "random_operands_clock_mhz": best[1][1],
                    "kernel_mix_tflops": best.get(7, (None, None))[0],
                    "frac_of_random_operands": round(roof["achieved"] / best[1][0], 4),
                    "source": os.path.relpath(path, ROOT)}
                return
        except (OSError, ValueError):
            continue


def measure(workload, args, rank, world, pairs=None, cloud_kind=None, skip_repeats=True, steps=None, warmup=None,
            precision=None):
    """one workload: timed region + (rank 0) roofline of its dominant launch; returns (record, state_dict)"""
    from pcr_amd import engine
    if precision is not None:
        with engine.precision(precision):
            return measure(workload, args, rank, world, pairs, cloud_kind, skip_repeats, steps, warmup)
    from pcr_amd import shard
    from pcr_amd import testing as T
    desc, kind, n, bl, dpairs = WORKLOADS[workload]
    pairs = pairs or dpairs
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    model, sd = build_model(kind, bl)
    if kind == "ssg":
        for sa in model.backbone.SA_modules:
            sa.skip_repeats = skip_repeats
    # weak scaling: every rank owns `pairs` independent pairs (its own seed), already resident in HBM
    cloud_kind = cloud_kind or ("box" if kind == "ssg" else "randn")   # ball-query radii are metric: box crops
    s1, s2 = T.synthetic_pairs(pairs, n, seed=1234 + rank, kind=cloud_kind)
    s1, s2 = s1.cuda(), s2.cuda()
    prewarm()
    # split-bf16 guard (pcr_amd/engine.py): the level these weights run at is calibrated on this batch, untimed (what
    # ReIDNet.forward_test does on its first batch; the WHOLE batch, so that every launch of the process has the timed
    # launches' size and the rocprofv3 per-kernel averages stay comparable with the event-timed ones); level and measured
    # deviations go into config.guard
    guard = (model.calibrate_precision(s1, s2, max_pairs=int(s1.shape[0]))
             if (engine.GUARD and engine.PRECISION == "bf16x3") else None)
    step_fn, launch_mode = graph_step(lambda: hot_path(model, s1, s2))
    pilot_ms = {k: round(v, 4) for k, v in LAST_PILOT.items()}
    with torch.no_grad():
        dt, out = shard.timed(step_fn, steps, warmup, sync=torch.cuda.synchronize, device="cuda")
    assert torch.isfinite(out).all()
    out = out.clone()
    del step_fn
    per_rank = [t / steps * 1e3 for t in shard.LAST_RANK_SECONDS]      # every rank's own time (the max is ms_per_step)
    rec = None
    if rank == 0:
        clk = clock_probe()
        roof, _ = roofline_of(model, s1, s2, workload, pairs)
        add_clock(roof, clk)
        add_sustained(roof)
        with torch.no_grad(), engine.precision("f32"):
            ref = hot_path(model, s1, s2)
        rec = dict(value=world * pairs * steps / dt, unit="pairs/s", steps=steps, warmup=warmup,
                   ms_per_step=dt / steps * 1e3, per_rank_ms_per_step=per_rank,
                   # the arithmetic the workload's dominant matrix launch ran in (PointNet / DGCNN: their encoder GEMMs
                   # are f32-input MFMA whatever the mode; only their matching attention follows it)
                   dtype=(roof.get("kernel_arithmetic") if roof["bound"] == "mfma" and roof.get("kernel_arithmetic")
                          else PREC_INFO[engine.PRECISION][0]),
                   max_abs_dlogit_vs_f32_path=float((out - ref).abs().max()),
                   data="synthetic (%s clouds, seeded random-init weights with non-trivial BN statistics)" % cloud_kind,
                   config={"workload": "%s: %s" % (workload, desc), "pairs_per_gpu_per_step": pairs, "points": n,
                           "backbone_list": bl, "parallelism": "independent pair shards x%d" % world,
                           "rccl_ranks": world,
                           "precision": precision_text(engine.PRECISION), "precision_tag": engine.PRECISION,
                           "launch": launch_mode, **pilot_ms},
                   roofline=roof)
        if guard is not None:
            rec["config"]["guard"] = {"level": guard["level"], "bound": guard["bound"],
                                      "dlogit": {str(k): float("%.2e" % v) for k, v in guard["dlogit"].items()}}
            sen = sentinel_cost(model, s1, s2, rec["ms_per_step"])
            if sen:
                rec["config"]["guard"]["sentinel"] = sen
        if kind == "ssg":
            rec["config"]["fill"] = ssg_fill(model, s1)
            rec["config"]["skip_repeats"] = bool(skip_repeats)
        else:
            rec["config"]["fill"] = "kNN groups: every one of the K rows is a genuine neighbour (fill 1.0)"
        if args.detail:
            det = profile_kernels(model, s1, s2, detail=True)
            for k, v in sorted(det.items(), key=lambda kv: -kv[1][0]):
                print("%-52s %8.3f ms x%d  %7.2f TFLOP/s  %7.1f GB/s(alg)" % (
                    k, v[0], v[1], v[2] / (v[0] * 1e-3) / 1e12, v[3] / (v[0] * 1e-3) / 1e9), file=sys.stderr)
    del model, s1, s2
    torch.cuda.empty_cache()
    return rec, sd


def sentinel_cost(model, s1, s2, ms_per_step, reps=3):
    """what the run-time sentinel of the split-bf16 guard costs an EAGER caller (a replayed graph contains no check): one
    check = the f32 path + the current level on engine.GUARD_SENTINEL_PAIRS pairs of the live batch, every
    engine.GUARD_EVERY batches -> overhead = check / (every x step)"""
    from pcr_amd import engine
    st = model.__dict__.get("_pcr_guard")
    if st is None or engine.GUARD_EVERY <= 0:
        return None
    model._sentinel(s1, s2, st)                     # (plans of the sample's shape)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model._sentinel(s1, s2, st)
    torch.cuda.synchronize()
    check_ms = (time.perf_counter() - t0) / reps * 1e3
    return {"every": engine.GUARD_EVERY, "pairs": min(engine.GUARD_SENTINEL_PAIRS, int(s1.shape[0])),
            "check_ms": round(check_ms, 3), "overhead": round(check_ms / (engine.GUARD_EVERY * ms_per_step), 5),
            "worst": float("%.2e" % st["sentinel"]["worst"]), "level": st["level"]}


def gallery_bench(args, desc, n, bl, pairs, rank, world, steps=None, warmup=None, cpu=True):
    """G tracks x G detections; encode the 2G objects once, score all G*G combinations"""
    from pcr_amd import engine, shard
    from pcr_amd import testing as T
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    G = max(1, int(round(pairs ** 0.5)))
    P = G * G
    model, sd = build_pt_model(bl)
    clouds = T.synthetic_clouds(2 * G, n, seed=1234 + rank, kind="randn").cuda()
    ii, jj = torch.meshgrid(torch.arange(G), torch.arange(G, 2 * G), indexing="ij")
    combos = torch.stack([ii.reshape(-1), jj.reshape(-1)], dim=1).cuda()

    def step():
        xyz, h = model.forward_inference(clouds)
        return model.match_gallery(h, xyz, combos)
    prewarm()
    guard = model.calibrate_precision(clouds[:G], clouds[G:]) if (engine.GUARD and engine.PRECISION == "bf16x3") else None
    step_fn, launch_mode = graph_step(step)
    pilot_ms = {k: round(v, 4) for k, v in LAST_PILOT.items()}
    with torch.no_grad():
        dt, out = shard.timed(step_fn, steps, warmup, sync=torch.cuda.synchronize, device="cuda")
    assert torch.isfinite(out).all() and out.numel() == P
    del step_fn
    line = None
    if rank == 0:
        clk = clock_probe()
        roof, _ = roofline_of(None, None, None, "gallery128", P, fn=step)
        add_clock(roof, clk)
        line = {"metric": "siamese pair-comparisons/sec @%d pts (gallery: every object encoded once)" % n,
                "value": world * P * steps / dt, "unit": "pairs/s", "n_gpus": world, "steps": steps,
                "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": PREC_INFO[engine.PRECISION][0],
                "data": "synthetic (randn clouds, seeded random-init weights with non-trivial BN statistics)",
                "config": {"workload": "gallery128: %s" % desc, "precision": precision_text(engine.PRECISION), "pairs_per_gpu_per_step": P, "objects_per_gpu_per_step": 2 * G,
                           "points": n, "backbone_list": bl, "parallelism": "independent galleries x%d" % world,
                           "rccl_ranks": world, "launch": launch_mode, "precision_tag": engine.PRECISION, **pilot_ms},
                "roofline": roof}
        if guard is not None:
            line["config"]["guard"] = {"level": guard["level"], "bound": guard["bound"],
                                       "dlogit": {str(k): float("%.2e" % v) for k, v in guard["dlogit"].items()}}
        if cpu and world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = gallery_cpu_baseline(sd, n, bl, G)
    del model, clouds, combos
    torch.cuda.empty_cache()
    return line


COMPACT_LIMIT = 8000        # bytes of the LAST stdout line (the driver's parser lost a 27 KB line in round 4; target <= 4 KB)
FULL_RECORD = "bench_full.json"


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def compact_roofline(roof):
    """the contract's roofline object (bound / achieved / peak / unit / frac / traffic) plus the few figures needed to
    re-derive it; the prose (`clock_source`, `traffic_source`) and the long per-launch maps stay in the full record"""
    if not roof:
        return roof
    keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches_per_step",
            "kernel_arithmetic", "mfma_per_product", "issued_gflop_per_launch", "algorithmic_mb_per_launch",
            "algorithmic_gflop_per_launch", "mfma_pipe_busy_pmc", "share_of_step", "clock_ghz", "valu_issue", "sustained_bf16")
    out = {k: _r(roof[k]) for k in keep if k in roof}
    out.setdefault("traffic", None)
    per = roof.get("per_kernel_ms") or roof.get("profiled_kernels_ms")
    if per:
        out["per_kernel_ms"] = {k: _r(v, 3) for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:10]}
    return out


def compact_config(cfg):
    keep = ("pairs_per_gpu_per_step", "objects_per_gpu_per_step", "points", "rccl_ranks", "skip_repeats", "eager_ms",
            "hipgraph_ms", "guard")
    out = {"workload": str(cfg.get("workload", ""))[:110]}
    out.update({k: _r(cfg[k], 3) for k in keep if k in cfg})
    if "launch" in cfg:
        out["launch"] = str(cfg["launch"]).split(" ")[0].rstrip(":")
    if "precision_tag" in cfg:
        out["precision"] = cfg["precision_tag"]
    if isinstance(cfg.get("fill"), dict):
        out["fill"] = {k: _r(v["fill"], 3) for k, v in cfg["fill"].items()}
    if "parallelism" in cfg:
        out["parallelism"] = str(cfg["parallelism"])[:80]
    return out


def compact_also(rec):
    if "error" in rec:
        return {"name": rec.get("name"), "error": str(rec["error"])[:160]}
    roof = rec.get("roofline") or {}
    out = {"name": rec.get("name"), "value": _r(rec.get("value"), 1),
           "ms_per_step": _r(rec.get("ms_per_step"), 3), "dtype": str(rec.get("dtype")).split(";")[0][:16],
           "pairs": (rec.get("config") or {}).get("pairs_per_gpu_per_step"),
           "kernel": str(roof.get("kernel", "")).split("[")[0], "bound": roof.get("bound"), "frac": _r(roof.get("frac"), 3)}
    if rec.get("n_gpus", 1) != 1:
        out["n_gpus"] = rec["n_gpus"]
    if rec.get("unit") not in (None, "pairs/s"):
        out["unit"] = rec["unit"]
    if "mfma_pipe_busy_pmc" in roof:
        out["pipe_busy_pmc"] = _r(roof["mfma_pipe_busy_pmc"], 3)
    if "max_abs_dlogit_vs_f32_path" in rec:
        out["dlogit_vs_f32"] = float("%.2e" % rec["max_abs_dlogit_vs_f32_path"])
    return out


def compact_line(full):
    """full record (everything measured) -> the ONE line the contract asks for, a few KB: headline fields, the headline's
    roofline, cpu_baseline, companions as {name, value, unit, ms_per_step, dtype, frac, ..}.  Pure (CPU-tested)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
    line = {k: full[k] for k in keep if k in full}
    line["data"] = str(line.get("data", ""))[:110]
    line["config"] = compact_config(full.get("config", {}))
    line["roofline"] = compact_roofline(full.get("roofline"))
    if "cpu_baseline" in full:
        cb = dict(full["cpu_baseline"])
        cb["sample"] = str(cb.get("sample", ""))[:80]
        cb["value"] = _r(cb.get("value"), 3)
        line["cpu_baseline"] = cb
    if "max_abs_dlogit_vs_f32_path" in full:
        line["max_abs_dlogit_vs_f32_path"] = full["max_abs_dlogit_vs_f32_path"]
    if "per_rank_ms_per_step" in full and len(full["per_rank_ms_per_step"]) > 1:
        line["per_rank_ms_per_step"] = list(full["per_rank_ms_per_step"])
    if full.get("also"):
        line["also"] = [compact_also(a) for a in full["also"]]
    line["full_record"] = FULL_RECORD
    return line


def emit(full, out=None, err=None, root=None):
    """the FULL record goes to bench_full.json under `root` (default: beside this file) and to stderr, one line, prefixed;
    the LAST stdout line is the compact record.  Nothing is printed after it, and SOMETHING parseable always is: a record
    the compaction chokes on still yields {metric, value, unit, ...}."""
    out = out or sys.stdout
    err = err or sys.stderr
    try:
        blob = json.dumps(full)
        try:
            with open(os.path.join(root or ROOT, FULL_RECORD), "w") as f:
                f.write(blob + "\n")
        except OSError:
            pass
        err.write("bench full record: " + blob + "\n")
        err.flush()
    except (TypeError, ValueError) as e:
        err.write("bench full record: not serialisable (%s)\n" % e)
    try:
        line = json.dumps(compact_line(full))
        if len(line) >= COMPACT_LIMIT:              # never lose the headline to a parser again: drop the companions first
            slim = compact_line(full)
            slim["also"] = [{"name": a.get("name"), "value": a.get("value"), "ms_per_step": a.get("ms_per_step")}
                            for a in slim.get("also", [])]
            if isinstance(slim.get("roofline"), dict):
                slim["roofline"].pop("per_kernel_ms", None)
            line = json.dumps(slim)
            if len(line) >= COMPACT_LIMIT:
                slim.pop("also", None)
                line = json.dumps(slim)
    except Exception as e:                          # (last resort: the contract's scalar fields only)
        keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype")
        mini = {k: full.get(k) for k in keep if isinstance(full.get(k), (int, float, str, bool, type(None)))}
        mini["compaction_error"] = "%s: %s" % (type(e).__name__, str(e)[:120])
        line = json.dumps(mini)
    out.write(line + "\n")
    out.flush()


def finish(line, rank):
    """rank 0 prints the ONE JSON line; every rank leaves the process group"""
    from pcr_amd import shard
    if rank == 0 and line is not None:
        emit(line)
    if shard.is_dist():
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def gallery_cpu_baseline(sd, n, bl, G, budget_s=20.0):
    """the torch restatement on the host: encode 8 objects, match 16 combinations, project to the job's shape
    (2G encodes + G*G matches) -- the CPU cannot run 36 k comparisons in a bench's time"""
    import time
    import model_oracle as MO
    from pcr_amd import testing as T
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(32, avail))
    torch.set_num_threads(threads)
    clouds = T.synthetic_clouds(8, n, seed=1234, kind="randn")
    pairs = torch.tensor([[i, (i + 1 + k) % 8] for i in range(8) for k in range(2)])
    with torch.no_grad():
        t0 = time.perf_counter()
        _, h = MO.pt_backbone(MO._sub(sd, "backbone."), clouds, bl)
        t_enc = (time.perf_counter() - t0) / 8
        t0 = time.perf_counter()
        MO.match(sd, h[pairs[:, 0]], clouds[pairs[:, 0]], h[pairs[:, 1]], clouds[pairs[:, 1]])
        t_match = (time.perf_counter() - t0) / len(pairs)
    P = G * G
    return {"value": P / (2 * G * t_enc + P * t_match), "unit": "pairs/s", "cores": threads, "threads_used": threads,
            "host_cores": avail, "kind": "port",
            "sample": "8 objects encoded + 16 combinations matched by the torch eager fp32 restatement "
                      "(oracle/model_oracle.py), projected to 2G = %d encodes + G*G = %d matches per step" % (2 * G, P)}


def launch_ranks(args, argv):
    """`--gpus N` without a torchrun environment: this process has not touched the GPU (nothing here calls into HIP
    before this point) and never will -- it starts N fresh ranks with torch.distributed.run as CHILD processes,
    relays their output (rank 0 prints the JSON line) and exits with their status.  (Reference counterpart:
    `torchpack dist-run -np N`, launcher_training.py:65 / tools/train.py:26.)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in proc.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
    raise SystemExit(proc.wait())


def dry_run(args):
    """CPU / gloo rehearsal of the launch + timing protocol (tests/test_distributed.py): same rendezvous, barrier
    bracket, max-reduce and JSON line; the step is a stand-in, no throughput is claimed."""
    from pcr_amd import shard
    rank, local, world = shard.init(backend="gloo")
    x = torch.ones(64, 64)
    dt, out = shard.timed(lambda: (x @ x).sum(), args.steps, args.warmup)
    if rank == 0:
        print(json.dumps({"metric": "dry-run (no GPU work)", "value": 0.0, "unit": "pairs/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
                          "data": "none", "config": {"workload": "dry-run", "rccl_ranks": world,
                                                     "backend": "gloo"}}), flush=True)
    if shard.is_dist():
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pt1024", choices=sorted(WORKLOADS),
                    help="default = pt1024, BASELINE.json configs[2]: the reference's own Point-Transformer config at the "
                         "metric's 1024 points (the model the reference has a config and a checkpoint for, golden-pinned "
                         "stage by stage); ssg1024 = configs[1], a composition with no reference config (companions)")
    ap.add_argument("--pairs", type=int, default=0, help="pairs per GPU per step (default: per workload)")
    ap.add_argument("--clouds", default=None, choices=["box", "dup", "crop", "randn"], help="synthetic cloud distribution")
    ap.add_argument("--full-groups", action="store_true",
                    help="SSG: evaluate all K rows of every ball-query group (no duplicate-row skipping)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the companion measurements of the default run")
    ap.add_argument("--detail", action="store_true", help="also print per-launch device times (stderr)")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo rehearsal of the launch + timing protocol")
    args = ap.parse_args()

    # ---- rank launch: BEFORE anything touches the GPU (torch.cuda.is_available() would) ----
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return launch_ranks(args, sys.argv[1:])
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks"
                         % (args.gpus, os.environ["WORLD_SIZE"]))
    if args.dry_run:
        return dry_run(args)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    from pcr_amd import shard
    rank, local, world = shard.env_world()
    # (test hooks, tests/test_gpu_distributed.py: RCCL refuses two ranks on one device, so the multi-rank code path is
    # exercised on a 1-GPU box with every rank on cuda:0 over gloo; the driver's runs use neither variable)
    if os.environ.get("PCR_BENCH_TEST_SAME_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    shard.init(backend=os.environ.get("PCR_BENCH_TEST_BACKEND", "nccl"),       # "nccl" is RCCL on ROCm
               device=torch.device("cuda", local))
    if shard.is_dist():
        import torch.distributed as dist
        assert dist.get_world_size() == args.gpus

    desc, kind, n, bl, dpairs = WORKLOADS[args.workload]
    if kind == "pt_train":
        return finish(train_bench(args, desc, n, bl, args.pairs or dpairs, rank, world), rank)
    if kind == "gallery":
        return finish(gallery_bench(args, desc, n, bl, args.pairs or dpairs, rank, world), rank)

    rec, sd = measure(args.workload, args, rank, world, pairs=args.pairs or None, cloud_kind=args.clouds,
                      skip_repeats=not args.full_groups)
    default_run = (args.workload == "pt1024" and not args.pairs and not args.clouds and not args.full_groups)
    also = []
    if default_run and world > 1 and not args.no_also:
        # N > 1: the inference headline has no collective at all, so a scaling run would never exercise the gradient
        # all-reduce north_star names (reference: mmdet3d/apis/train.py:35-56).  The training companion (BASELINE config 4:
        # forward + backward + ONE flat-bucket RCCL all-reduce + clip + AdamW) therefore runs on every rank here too.
        d, _, n_, bl_, p_ = WORKLOADS["pt128_train"]
        try:
            r = train_bench(args, d, n_, bl_, p_, rank, world)
        except Exception as e:
            r = {"error": "%s: %s" % (type(e).__name__, e)}
        if r is not None:
            r["name"] = "pt128_train"
            also.append(r)
    if default_run and world == 1 and not args.no_also:
        # Beside the headline (the reference's Point-Transformer config in split bf16, guarded), in the same run: the SAME
        # workload in the reference's own arithmetic (f32-input MFMA, guard not involved); BASELINE configs[4]'s shape in both
        # arithmetics; BASELINE configs[1] (the PointNet++ SSG composition -- no reference config, restatement-pinned) with
        # its variants: uniform box clouds leave its ball-query groups nearly empty (config.fill) and the ragged SA kernel
        # skips the repeated rows, so it also runs on clouds with 50 % duplicated points, on crops resampled to 1024 WITH
        # replacement (what the reference's subsamplePC hands the model) and with all K rows of every group evaluated;
        # BASELINE configs[0] (PointNet).
        for name, wl, kw in (("pt1024_f32", "pt1024", dict(precision="f32", steps=max(4, args.steps // 2))),
                             ("pt4096", "pt4096", dict(steps=max(4, args.steps // 2))),
                             ("pt4096_f32", "pt4096", dict(precision="f32", steps=max(3, args.steps // 5))),
                             ("ssg1024", "ssg1024", dict()),
                             ("ssg1024_b2048", "ssg1024", dict(pairs=2048)),     # (the batch of rounds 3-4: comparable across rounds)
                             ("ssg1024_f32", "ssg1024", dict(precision="f32")),
                             ("ssg1024_bf16", "ssg1024", dict(precision="bf16")),
                             ("ssg1024_dup", "ssg1024", dict(cloud_kind="dup")),
                             ("ssg1024_crop", "ssg1024", dict(cloud_kind="crop")),
                             ("ssg1024_full", "ssg1024", dict(skip_repeats=False, steps=max(4, args.steps // 4))),
                             ("pointnet256", "pointnet256", dict())):
            try:
                r, _ = measure(wl, args, rank, world, **kw)
                r["metric"] = "siamese pair-comparisons/sec @%d pts" % WORKLOADS[wl][2]
            except Exception as e:      # a companion must never cost the headline line
                r = {"error": "%s: %s" % (type(e).__name__, e)}
            r["name"] = name
            also.append(r)
        # BASELINE config 4 (the training step: forward + backward + bucket exchange + clip + AdamW) and SURVEY 8f rank 1
        # (gallery matching), each with the roofline of its own dominant launch
        for name, fn in (("pt128_train", train_bench), ("gallery128", gallery_bench)):
            d, _, n_, bl_, p_ = WORKLOADS[name]
            try:
                kw = dict(cpu=False) if name == "gallery128" else {}
                r = fn(args, d, n_, bl_, p_, rank, world, **kw)
            except Exception as e:
                r = {"error": "%s: %s" % (type(e).__name__, e)}
            r["name"] = name
            also.append(r)
    if rank == 0:
        line = {
            "metric": "siamese pair-comparisons/sec @%d pts" % n,
            "value": rec["value"], "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"], "per_rank_ms_per_step": rec["per_rank_ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": rec["dtype"], "max_abs_dlogit_vs_f32_path": rec["max_abs_dlogit_vs_f32_path"],
            "data": rec["data"], "config": rec["config"], "roofline": rec["roofline"],
        }
        if also:
            line["also"] = also
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, sd)
        emit(line)
    if shard.is_dist():
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
