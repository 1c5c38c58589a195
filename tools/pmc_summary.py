"""Reduces rocprofv3 --pmc passes of `bench.py` to the per-kernel JSON that bench.py reads for `roofline.traffic`
and `mfma_pipe_busy_pmc` (profiles/rNN_<workload>_pmc.json).

    python tools/pmc_summary.py OUT.json PAIRS_PER_STEP DIR_BUSY DIR_FETCH DIR_WRITE [PRECISION]

DIR_BUSY : --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE      (with --kernel-trace)
DIR_FETCH: --pmc FETCH_SIZE        DIR_WRITE: --pmc WRITE_SIZE                 (separate passes: TCC slots)
PRECISION: the arithmetic mode the profiled run was timed in (engine.PRECISION; default bf16x3).
Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request on gfx950 -> doubled;
WRITE_SIZE is exact.  GRBM_GUI_ACTIVE arrives summed over the 8 XCDs, so one XCD's cycles are GUI / 8 and
mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES (summed over all SIMDs) / (1024 SIMDs x GUI / 8).

Launch -> kernel map (`_launch_to_kernel`): a bench launch name is mapped to the kernel symbol that (a) matches one of the
launch's known prefixes, (b) carries the ARITHMETIC of the profiled mode in its template arguments (`arithmetic_of`;
symbols that exist in one arithmetic only are always eligible) and (c) among those was dispatched most often -- the
timed kernel runs warm-up + timed steps, a kernel of another arithmetic that the same process launches once (bench.py's
f32 comparison pass) never wins.  Round 3 picked max(launch_us) over a prefix that matched both arithmetics and quoted
the f32 one-shot's counters for the bf16x3 headline."""
import collections
import csv
import glob
import json
import re
import sys

ONE_FORM_F32 = ("attn_kv_kernel", "attn_kv_kernel_o3", "attn_kv_kernel_o2", "attn_kv_wide_kernel", "dense_kernel", "dense_gn_kernel", "dense_rw_kernel",
                "tdense_bwd_kernel", "tdense_fwd_kernel", "tstream_fwd_pipe_kernel", "tstream_fwd_kernel", "tstream_bwd_kernel")
ONE_FORM_BF3 = ("attn_kv_stream32_kernel", "attn_apply_stream64_kernel", "attn_kv_stream128_kernel",
                "attn_apply_stream128_kernel", "gallery_tail_kernel", "tdense_bwd_bf_kernel")


def short(name):
    m = re.search(r"(\w+_kernel\w*(<[^>]*>)?)", name)
    return m.group(1) if m else name[:48]


def targs(k):
    m = re.search(r"<([^>]*)>", k)
    return [a.strip() for a in m.group(1).split(",")] if m else []


def arithmetic_of(k):
    """arithmetic of a kernel symbol's matrix phases, from its template arguments (None: not known from the symbol)"""
    a = targs(k)
    base = k.split("<")[0]
    prec = {"0": "f32", "1": "bf16x3", "2": "bf16"}
    if base == "sa_rag_kernel" and len(a) >= 7:
        return prec.get(a[6])
    if base == "sa_fused_kernel" and len(a) >= 8:
        return prec.get(a[7])
    if base == "dense_pm_kernel" and len(a) >= 2:
        return prec.get(a[1])
    if base == "dense_pm_res_kernel" and len(a) >= 1:
        return "bf16x3" if a[0] == "3" else "bf16"
    if base in ("sa_stream_kernel", "sa_stream_rag_kernel", "sa_wsplit_rag_kernel") and len(a) >= 3:
        return "bf16x3" if a[2] == "true" else "bf16"
    if base == "attn_kv_stream64_kernel" and len(a) >= 2:
        return "bf16x3" if a[1] == "true" else "f32"
    if base in ("dense_bf_kernel", "dense_bf_pc_kernel") and len(a) >= 2:
        return "bf16x3" if a[1] == "3" else "bf16"
    if base == "dense_pm_stream_kernel" and len(a) >= 2:
        return "bf16x3" if a[1] == "true" else "bf16"
    if base in ONE_FORM_BF3:
        return "bf16x3"
    if base in ONE_FORM_F32:
        return "f32"
    return None          # e.g. attn_apply_kernel<TB, NR>: one symbol per unit, told apart by dispatch count only


def eligible(k, precision):
    base = k.split("<")[0]
    a = arithmetic_of(k)
    return a is None or a == precision or base in ONE_FORM_F32 or base in ONE_FORM_BF3


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
    return {k: (sum(v) / len(v) / 1e3, len(v)) for k, v in agg.items()}


# bench.py's launch names -> PREFIXES of candidate kernel symbols (template arguments after the prefix vary by build);
# the first prefix with an eligible hit wins
KNOWN = {
    "sa_ragged[D=128,c=128/128/256,N=512,S=128,K=64]": ["sa_wsplit_rag_kernel<", "sa_rag_kernel<2, 2, 1, 1, 1, 1"],
    "sa_ragged[D=0,c=64/64/128,N=1024,S=512,K=32]": ["sa_stream_rag_kernel<2, 4", "sa_rag_kernel<4, 1, 2, 1, 1, 2"],
    "sa_tables[D=128,out=128,N=512]": ["dense_pm_res_kernel<", "dense_pm_kernel<1,"],
    "dense[cin=256,cout=64,L=128]": ["dense_pm_stream_kernel<2", "dense_kernel<2, false, false"],
    "fps[N=1024,M=512]": ["fps_pair_kernel<", "fps_wave_kernel<8"],
    "ball_query[N=1024,M=512,K=32]": ["ball_query_rows_kernel<16", "ball_query_reg_kernel<16"],
    # pt1024: the K-row SA kernels (SA3 / SA2 / SA1) and the neighbour search
    "sa_fused[D=64,c=128/128/128,N=512,S=256,K=48]": ["sa_stream_kernel<4, 4", "sa_fused_kernel<3, 1, 1, 1, true, 1, 0"],
    "sa_fused[D=32,c=64/64/64,N=1024,S=512,K=48]": ["sa_stream_kernel<2, 2", "sa_fused_kernel<6, 1, 2, 2, true, 1, 0"],
    "sa_fused[D=0,c=32/32/32,N=1024,S=1024,K=32]": ["sa_stream_kernel<1, 1", "sa_fused_kernel<4, 1, 4, 4, true, 1,"],
    "attn_apply[d=64,c1=64,out=64,Lq=1024]": ["attn_apply_stream64_kernel<true, 4, 0", "attn_apply_kernel<2, 1>"],
    "attn_kv[d=64,c2=64,Sk=1024]": ["attn_kv_stream64_kernel<true, true, 4", "attn_kv_stream64_kernel<true, false, 4"],
    "attn_kv[d=128,c2=128,Sk=256]": ["attn_kv_stream128_kernel", "attn_kv_kernel_o2<1, 2, 1, 4", "attn_kv_kernel_o3<1, 2, 1, 4>"],
    "attn_kv[d=128,c2=128,Sk=1024]": ["attn_kv_kernel_o2<1, 2, 1, 4", "attn_kv_kernel_o3<1, 2, 1, 4>"],
    "attn_apply[d=128,c1=128,out=128,Lq=256]": ["attn_apply_stream128_kernel", "attn_apply_kernel<1, 2>"],
    "knn_prefix[N=1024,S=1024,K=32]": ["knn_prefix_reg_kernel<8>"],
    "knn_prefix2[N=1024,S=1024,K=32,S2=512,K2=48]": ["knn_prefix_reg_kernel<8>"],
    # pt4096 / gallery128 / pointnet256
    "knn_prefix[N=4096,S=4096,K=32]": ["knn_prefix_lds_kernel"],
    "knn_prefix2[N=4096,S=4096,K=32,S2=2048,K2=48]": ["knn_prefix_lds_kernel"],
    "sa_fused[D=64,c=128/128/128,N=2048,S=1024,K=48]": ["sa_stream_kernel<4, 4"],
    "attn_apply[d=64,c1=64,out=64,Lq=128]": ["gallery_tail_kernel", "attn_apply_stream64_kernel<false, 4, 0, 2, 2, false",
                                             "attn_apply_stream64_kernel<false, 4, 0, 2", "attn_apply_kernel<2, 1>"],
    # round 6 (ABI 16): the gallery's stage-2 launch with pooled output
    "attn_apply[d=64,c1=64,out=64,Lq=128,pooled]": ["attn_apply_stream64_kernel<false, 4, 0, 2, 2, true"],
    "attn_kv[d=64,c2=64,Sk=128]": ["attn_kv_stream64_kernel<true, true, 4", "attn_kv_stream64_kernel<true, false, 4"],
    "dense_gn[cin=1024,cout=512,L=256]": ["dense_bf_pc_kernel<true, 3, 8", "dense_bf_pc_kernel<true", "dense_bf_kernel<true, 3, 2",
                                          "dense_bf_kernel<true", "dense_kernel<2, true, true"],
    "dense_max[cin=128,cout=1024,L=256]": ["dense_rw_kernel<16, true", "dense_max_kernel"],
    "dense[cin=128,cout=1024,L=256]": ["dense_rw_kernel<16, false", "dense_kernel<2, false, false"],
    "dense[cin=512,cout=1024,L=256]": ["dense_bf_kernel<false, 3, 2", "dense_kernel<2, false, true"],
    "tdense_bwd[mode=1,cin=128,cout=128,L=1536]": ["tdense_bwd_bf_kernel", "tdense_bwd_kernel<1, 1, 4, 0, 0>"],
    "tdense_bwd[mode=3,cin=128,cout=128,L=1536]": ["tdense_bwd_bf_kernel", "tdense_bwd_kernel<1, 1, 4, 0, 0>"],
}


def map_launches(res, precision, known=KNOWN):
    """launch name -> kernel symbol (see the module docstring); `res`: {symbol: {"launch_us", "calls", ...}}"""
    l2k = {}
    for launch, prefixes in known.items():
        for prefix in prefixes:
            hits = [k for k in res if isinstance(res[k], dict) and k.startswith(prefix) and eligible(k, precision)]
            if hits:
                l2k[launch] = max(hits, key=lambda k: (res[k].get("calls", 0), res[k]["launch_us"]))
                break
    return l2k


def remap(path):
    """python tools/pmc_summary.py --remap FILE: recompute the arithmetic labels and the launch -> kernel map of an existing
    summary with THIS file's tables (kernels added after the counters were collected)"""
    res = json.load(open(path))
    precision = res.get("_precision", "bf16x3")
    for k, v in res.items():
        if isinstance(v, dict) and "launch_us" in v:
            v["arithmetic"] = arithmetic_of(k)
    l2k = map_launches(res, precision)
    for launch, k in l2k.items():
        if res[k]["arithmetic"] is None and launch.startswith("attn_apply"):
            res[k]["arithmetic"] = "f32" if precision == "f32" else "bf16x3"
    res["_launch_to_kernel"] = l2k
    json.dump(res, open(path, "w"), indent=1)
    print(json.dumps(l2k, indent=1))


def main():
    if sys.argv[1] == "--remap":
        return remap(sys.argv[2])
    out, pairs, d_busy, d_fetch, d_write = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    precision = sys.argv[6] if len(sys.argv) > 6 else "bf16x3"
    busy, fetch, write, dur = counters(d_busy), counters(d_fetch), counters(d_write), durations(d_busy)
    res = {}
    for k, c in busy.items():
        if "kernel" not in k or k.startswith("at::") or "elementwise" in k or "reduce_kernel" in k:
            continue
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        us, calls = dur.get(k, (0.0, 0))
        f_kb = fetch.get(k, {}).get("FETCH_SIZE", 0.0)
        w_kb = write.get(k, {}).get("WRITE_SIZE", 0.0)
        gui /= 8.0
        res[k] = {"launch_us": us, "calls": calls, "arithmetic": arithmetic_of(k),
                  "clock_ghz": gui / us / 1e3 if us else 0.0,
                  "mfma_pipe_busy": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui) if gui else 0.0,
                  "fetch_size_kb": f_kb, "write_size_kb": w_kb,
                  "hbm_bytes_corrected": (2.0 * f_kb + w_kb) * 1024.0}
    l2k = map_launches(res, precision)
    for launch, k in l2k.items():      # a symbol that does not name its arithmetic takes the mode's (dispatch count chose it)
        if res[k]["arithmetic"] is None and launch.startswith("attn_apply"):
            res[k]["arithmetic"] = "f32" if precision == "f32" else "bf16x3"
    res["_pairs_per_step"] = pairs
    res["_precision"] = precision
    res["_launch_to_kernel"] = l2k
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["launch_us"] if isinstance(kv[1], dict) and "launch_us" in kv[1] else 0):
        if isinstance(v, dict) and "launch_us" in v:
            print("%-44s %9.1f us x%-4d mfma busy %.2f  hbm %.1f MB" % (k, v["launch_us"], v["calls"], v["mfma_pipe_busy"], v["hbm_bytes_corrected"] / 1e6))


if __name__ == "__main__":
    main()
