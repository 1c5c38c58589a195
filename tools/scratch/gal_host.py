# where does the gallery128 step spend host time?  (host wall time of each call without sync = blocking time)
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench
from pcr_amd import testing as T
G = 192
model, sd = bench.build_pt_model([128, 64, 32])
clouds = T.synthetic_clouds(2 * G, 128, seed=1234, kind="randn").cuda()
ii, jj = torch.meshgrid(torch.arange(G), torch.arange(G, 2 * G), indexing="ij")
combos = torch.stack([ii.reshape(-1), jj.reshape(-1)], dim=1).cuda()
sync = torch.cuda.synchronize
with torch.no_grad():
    for _ in range(3):
        xyz, h = model.forward_inference(clouds); out = model.match_gallery(h, xyz, combos)
    sync()
    for rep in range(3):
        t0 = time.perf_counter(); xyz, h = model.forward_inference(clouds); t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
        out = model.match_gallery(h, xyz, combos); t3 = time.perf_counter(); sync(); t4 = time.perf_counter()
        print("encode host %.2f ms (+sync %.2f) | match host %.2f ms (+sync %.2f)" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
    # inside match_gallery: per call host time
    import pcr_amd.engine as E
    orig_apply = E.AttnPlan.apply; orig_kv = E.AttnPlan.kv
    def wrap(name, f):
        def g(*a, **k):
            t = time.perf_counter(); r = f(*a, **k); th = time.perf_counter() - t; sync(); tt = time.perf_counter() - t
            print("   %s host %.2f ms, with sync %.2f ms" % (name, th*1e3, tt*1e3)); return r
        return g
    E.AttnPlan.apply = wrap("apply", orig_apply); E.AttnPlan.kv = wrap("kv", orig_kv)
    sync(); t = time.perf_counter(); out = model.match_gallery(h, xyz, combos); sync(); print("match total (with per-call syncs) %.2f ms" % ((time.perf_counter()-t)*1e3))
    print("peak mem GB", torch.cuda.max_memory_allocated()/2**30, "reserved", torch.cuda.memory_reserved()/2**30)
    E.AttnPlan.apply = orig_apply; E.AttnPlan.kv = orig_kv
    ts = []
    for rep in range(24):
        sync(); t = time.perf_counter()
        xyz, h = model.forward_inference(clouds); out = model.match_gallery(h, xyz, combos)
        sync(); ts.append((time.perf_counter() - t) * 1e3)
    print("per-step ms:", " ".join("%.1f" % v for v in ts))
    st = torch.cuda.memory_stats()
    print("num_alloc_retries", st.get("num_alloc_retries"), "segments", st.get("segment.all.allocated"), "device allocs", st.get("num_device_alloc"), "device frees", st.get("num_device_free"))
    import gc
    print("gc enabled", gc.isenabled(), gc.get_threshold(), gc.get_count())
    E.AttnPlan.apply = wrap("apply", orig_apply); E.AttnPlan.kv = wrap("kv", orig_kv)
    head = model._head(h.device); orig_run = head.run.__func__ if hasattr(head.run, "__func__") else None
    for rep in range(6):
        sync(); t = time.perf_counter()
        xyz, h = model.forward_inference(clouds); sync(); te = time.perf_counter()
        out = model.match_gallery(h, xyz, combos)
        sync(); print("STEP %d: encode %.1f total %.1f ms" % (rep, (te - t) * 1e3, (time.perf_counter() - t) * 1e3))
    gc.disable()
    E.AttnPlan.apply = orig_apply; E.AttnPlan.kv = orig_kv
    ts = []
    for rep in range(12):
        sync(); t = time.perf_counter()
        xyz, h = model.forward_inference(clouds); out = model.match_gallery(h, xyz, combos)
        sync(); ts.append((time.perf_counter() - t) * 1e3)
    print("gc disabled per-step ms:", " ".join("%.1f" % v for v in ts))
