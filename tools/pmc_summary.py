"""Reduces rocprofv3 --pmc passes of `bench.py` to the per-kernel JSON that bench.py reads for `roofline.traffic`
and `mfma_pipe_busy_pmc` (profiles/rNN_<workload>_pmc.json).

    python tools/pmc_summary.py OUT.json PAIRS_PER_STEP DIR_BUSY DIR_FETCH DIR_WRITE

DIR_BUSY : --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE      (with --kernel-trace)
DIR_FETCH: --pmc FETCH_SIZE        DIR_WRITE: --pmc WRITE_SIZE                 (separate passes: TCC slots)
Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request on gfx950 -> doubled;
WRITE_SIZE is exact.  GRBM_GUI_ACTIVE arrives summed over the 8 XCDs, so one XCD's cycles are GUI / 8 and
mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES (summed over all SIMDs) / (1024 SIMDs x GUI / 8)."""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"(\w+_kernel\w*(<[^>]*>)?)", name)
    return m.group(1) if m else name[:48]


def counters(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


def durations(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return {k: sum(v) / len(v) / 1e3 for k, v in agg.items()}


def main():
    out, pairs, d_busy, d_fetch, d_write = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    busy, fetch, write, dur = counters(d_busy), counters(d_fetch), counters(d_write), durations(d_busy)
    res = {}
    for k, c in busy.items():
        if "kernel" not in k or k.startswith("at::") or "elementwise" in k or "reduce_kernel" in k:
            continue
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        us = dur.get(k, 0.0)
        f_kb = fetch.get(k, {}).get("FETCH_SIZE", 0.0)
        w_kb = write.get(k, {}).get("WRITE_SIZE", 0.0)
        gui /= 8.0
        res[k] = {"launch_us": us, "clock_ghz": gui / us / 1e3 if us else 0.0,
                  "mfma_pipe_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui) if gui else 0.0,
                  "fetch_size_kb": f_kb, "write_size_kb": w_kb,
                  "hbm_bytes_corrected": (2.0 * f_kb + w_kb) * 1024.0}
    res["_pairs_per_step"] = pairs
    # bench.py's launch names -> kernel symbols (ssg1024: the launches whose roofline object quotes PMC figures)
    # (values are PREFIXES of the kernel symbol: later template arguments -- arithmetic, bf-image flag -- vary by build)
    # (several prefixes: the first one that occurs wins -- the wave-autonomous kernels of the default arithmetic before
    # the tile kernels, which the same run also launches once for its f32 comparison pass)
    known = {"sa_ragged[D=128,c=128/128/256,N=512,S=128,K=64]": ["sa_rag_kernel<2, 2, 1, 1, 1, 1"],
             "sa_ragged[D=0,c=64/64/128,N=1024,S=512,K=32]": ["sa_stream_rag_kernel<2, 4", "sa_rag_kernel<4, 1, 2, 1, 1, 2"],
             "sa_tables[D=128,out=128,N=512]": ["dense_pm_kernel<1>"], "fps[N=1024,M=512]": ["fps_wave_kernel<8>"],
             "ball_query[N=1024,M=512,K=32]": ["ball_query_reg_kernel<16"],
             # pt1024: the K-row SA kernels (SA3 / SA2 / SA1) and the neighbour search
             "sa_fused[D=64,c=128/128/128,N=512,S=256,K=48]": ["sa_stream_kernel<4, 4", "sa_fused_kernel<3, 1, 1, 1, true, 1, 0"],
             "sa_fused[D=32,c=64/64/64,N=1024,S=512,K=48]": ["sa_stream_kernel<2, 2", "sa_fused_kernel<6, 1, 2, 2, true, 1, 0"],
             "sa_fused[D=0,c=32/32/32,N=1024,S=1024,K=32]": ["sa_stream_kernel<1, 1", "sa_fused_kernel<4, 1, 4, 4, true, 1,"],
             "attn_apply[d=64,c1=64,out=64,Lq=1024]": ["attn_apply_stream64_kernel<true, 4, false"],
             "attn_kv[d=64,c2=64,Sk=1024]": ["attn_kv_stream64_kernel<true, true"],
             "knn_prefix[N=1024,S=1024,K=32]": ["knn_prefix_reg_kernel<8>"]}
    l2k = {}
    for launch, prefixes in known.items():
        for prefix in prefixes:
            hits = [k for k in res if isinstance(res[k], dict) and k.startswith(prefix)]
            if hits:
                l2k[launch] = max(hits, key=lambda k: res[k]["launch_us"])
                break
    res["_launch_to_kernel"] = l2k
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["launch_us"] if isinstance(kv[1], dict) and "launch_us" in kv[1] else 0):
        if isinstance(v, dict) and "launch_us" in v:
            print("%-44s %9.1f us  mfma busy %.2f  hbm %.1f MB" % (k, v["launch_us"], v["mfma_pipe_busy"], v["hbm_bytes_corrected"] / 1e6))


if __name__ == "__main__":
    main()
