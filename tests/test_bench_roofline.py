"""bench.py's roofline object and tools/pmc_summary.py's launch -> kernel map on synthetic inputs (CPU): the label and the
peak follow the arithmetic the LAUNCH ran (never the mode asked for), no fraction above 1 from a mislabelled peak, and the
PMC figures quoted beside a launch come from the kernel of the same arithmetic that the timed steps dispatched
(a frac of 2.31, a pipe-busy figure of the f32 one-shot kernel, a stale dtype)."""
import importlib.util
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

_spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(ROOT, "tools", "pmc_summary.py"))
pmc_summary = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(pmc_summary)


def rec(name, ms, flops=0.0, nbytes=0.0, exec_flops=None, arith=None):
    return (name, 0.0, ms, flops, nbytes, flops if exec_flops is None else exec_flops, arith)


def agg(records):
    return bench.aggregate_profile(records, detail=True, elapsed=lambda e0, e1: e1 - e0)


# the round-3 gallery step: the dominant launch is a split-bf16 apply kernel issuing 3 x 328.6 GFLOP in 0.942 ms
GALLERY = [rec("attn_apply[d=64,c1=64,out=64,Lq=128]", 0.942, 328.6e9, 2.2e9, arith="bf16x3") for _ in range(4)] + \
          [rec("attn_kv[d=64,c2=64,Sk=128]", 0.52, 100e9, 0.8e9, arith="bf16x3"), rec("pool_head", 0.66, 1e9, 1.2e9)]


def test_gallery_launch_is_priced_in_the_arithmetic_it_ran():
    roof = bench.roofline_object(agg(GALLERY))
    assert roof["kernel"].startswith("attn_apply") and roof["kernel_arithmetic"] == "bf16x3"
    assert roof["peak"] == bench.MFMA_BF16_PEAK_TF and roof["mfma_per_product"] == 3
    assert 0.40 < roof["frac"] < 0.44          # the judge's recomputation: 1 047 TF = 0.42 (the r03 line said 2.31)
    assert roof["launches_per_step"] == 4


@pytest.mark.parametrize("arith,peak,mult", [("f32", bench.MFMA_F32_PEAK_TF, 1), ("bf16x3", bench.MFMA_BF16_PEAK_TF, 3),
                                              ("bf16", bench.MFMA_BF16_PEAK_TF, 1)])
def test_label_peak_and_multiplier_follow_the_launch(arith, peak, mult):
    # the same launch at the product rate of the f32 peak: a fraction of 1 in f32, 3/16 in split bf16, 1/16 in bf16
    gflop, ms = 100.0, 100.0 / bench.MFMA_F32_PEAK_TF
    roof = bench.roofline_object(agg([rec("sa_ragged[x]", ms, 10 * gflop * 1e9, 1e6, gflop * 1e9, arith),
                                      rec("fps[N=1024,M=512]", ms / 2, 0, 1e6)]))
    assert roof["kernel_arithmetic"] == arith and roof["peak"] == peak and roof["mfma_per_product"] == mult
    assert roof["frac"] == pytest.approx(mult * bench.MFMA_F32_PEAK_TF / peak, rel=1e-6)
    assert roof["frac"] <= 1.0 + 1e-9
    assert roof["product_frac_of_f32_mfma_peak"] == pytest.approx(1.0, rel=1e-6)


def test_launches_that_do_not_multiply_are_priced_against_hbm():
    roof = bench.roofline_object(agg([rec("knn_prefix[N=4096,S=4096,K=32]", 6.6, 1e12, 0.3e9),
                                      rec("sa_fused[x]", 4.7, 3e12, 1e8, 2.4e12, "bf16x3")]))
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == bench.HBM_PEAK_GBS
    assert roof["frac"] == pytest.approx(0.3e9 / 6.6e-3 / 1e9 / 8000.0)


def test_every_mode_has_a_true_precision_text():
    for mode in ("f32", "bf16x3", "bf16"):
        txt = bench.precision_text(mode)
        assert txt.startswith(mode)
    assert "tables" in bench.precision_text("bf16x3") and "attention" in bench.precision_text("bf16x3")
    assert "f32-input MFMA" not in bench.precision_text("bf16x3").split(";")[0]


# ---- tools/pmc_summary.py -------------------------------------------------------------------------------------------
RES = {
    "sa_rag_kernel<2, 2, 1, 1, 1, 1, 1>": {"launch_us": 1907.7, "calls": 7, "mfma_pipe_busy": 0.363, "hbm_bytes_corrected": 1.877e9},
    "sa_rag_kernel<2, 2, 1, 1, 1, 1, 0>": {"launch_us": 4292.8, "calls": 1, "mfma_pipe_busy": 0.718, "hbm_bytes_corrected": 1.875e9},
    "attn_kv_stream64_kernel<true, true, 4>": {"launch_us": 141.5, "calls": 14, "mfma_pipe_busy": 0.29, "hbm_bytes_corrected": 2e8},
    "attn_kv_stream64_kernel<true, false, 4>": {"launch_us": 211.0, "calls": 2, "mfma_pipe_busy": 0.60, "hbm_bytes_corrected": 2e8},
    "attn_apply_kernel<2, 1>": {"launch_us": 402.0, "calls": 2, "mfma_pipe_busy": 0.54, "hbm_bytes_corrected": 5e8},
    "attn_apply_stream64_kernel<false, 4, 0, 2, 2>": {"launch_us": 938.0, "calls": 28, "mfma_pipe_busy": 0.50, "hbm_bytes_corrected": 2.26e9},
}
SA2 = "sa_ragged[D=128,c=128/128/256,N=512,S=128,K=64]"


def test_pmc_map_picks_the_kernel_of_the_profiled_arithmetic():
    l2k = pmc_summary.map_launches(RES, "bf16x3")
    assert l2k[SA2] == "sa_rag_kernel<2, 2, 1, 1, 1, 1, 1>"          # round 3 mapped the slower f32 one-shot (0.718)
    assert l2k["attn_kv[d=64,c2=64,Sk=128]"] == "attn_kv_stream64_kernel<true, true, 4>"
    assert l2k["attn_apply[d=64,c1=64,out=64,Lq=128]"].startswith("attn_apply_stream64_kernel<false, 4, 0, 2")
    f32 = pmc_summary.map_launches(RES, "f32")
    assert f32[SA2] == "sa_rag_kernel<2, 2, 1, 1, 1, 1, 0>"
    assert f32["attn_kv[d=64,c2=64,Sk=128]"] == "attn_kv_stream64_kernel<true, false, 4>"
    assert pmc_summary.arithmetic_of("dense_pm_kernel<1, 1>") == "bf16x3"
    assert pmc_summary.arithmetic_of("sa_stream_rag_kernel<2, 4, true>") == "bf16x3"
    assert pmc_summary.arithmetic_of("sa_fused_kernel<3, 1, 1, 1, true, 1, 0, 0, false>") == "f32"


def test_bench_quotes_pmc_only_from_a_profile_of_the_same_mode_and_arithmetic(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    old = {"_launch_to_kernel": {SA2: "k0"}, "k0": {"launch_us": 4292.8, "mfma_pipe_busy": 0.718,
                                                           "hbm_bytes_corrected": 1.875e9}, "_pairs_per_step": 2048}
    (prof / "r03e_ssg1024_pmc.json").write_text(json.dumps(old))            # no _precision: a round-3 file
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    roof = bench.attach_pmc({"kernel": SA2, "kernel_arithmetic": "bf16x3", "traffic": None}, "ssg1024", 2048, "bf16x3")
    assert roof["traffic"] is None and "mfma_pipe_busy_pmc" not in roof
    new = {"_precision": "bf16x3", "_pairs_per_step": 2048, "_launch_to_kernel": {SA2: "k1"},
           "k1": {"launch_us": 1907.7, "mfma_pipe_busy": 0.363, "hbm_bytes_corrected": 1.877e9, "arithmetic": "bf16x3"}}
    (prof / "r04a_ssg1024_pmc.json").write_text(json.dumps(new))
    roof = bench.attach_pmc({"kernel": SA2, "kernel_arithmetic": "bf16x3", "traffic": None}, "ssg1024", 1024, "bf16x3")
    assert roof["mfma_pipe_busy_pmc"] == 0.363 and roof["traffic"] == pytest.approx(1.877e9 / 2)
    wrong = bench.attach_pmc({"kernel": SA2, "kernel_arithmetic": "f32", "traffic": None}, "ssg1024", 2048, "f32")
    assert wrong["traffic"] is None


def test_graph_step_falls_back_to_the_eager_callable(monkeypatch):
    """bench.graph_step is an optimisation of the measurement: without a device to capture on (this box), or with
    PCR_BENCH_GRAPH=0, it must hand back the callable it was given and say why; the mode is fixed by the environment,
    never by which way a pilot ran faster"""
    import inspect
    calls = []

    def fn():
        calls.append(1)
        import torch
        return torch.zeros(3)
    got, mode = bench.graph_step(fn)
    assert got is fn and mode.startswith("eager"), mode
    monkeypatch.setenv("PCR_BENCH_GRAPH", "0")
    got, mode = bench.graph_step(fn)
    assert got is fn and mode == "eager: PCR_BENCH_GRAPH=0"
    src = inspect.getsource(bench.graph_step)
    assert "te < tg" not in src and "want_graph" in src


# ---- the line the driver parses (a 27 KB line came back `parsed: null`) -----------------------------
def _full_record(n_also=12):
    """a default-run record the size of round 4's: long prose in every companion, per-kernel maps, PMC sources"""
    prose = bench.precision_text("bf16x3")
    per = {"kernel_%02d[D=128,c=128/128/256,N=512,S=128,K=64]" % i: 0.123456789 * (i + 1) for i in range(14)}

    def roof(kernel):
        return {"kernel": kernel, "bound": "mfma", "achieved": 884.123456789, "peak": 2500.0, "unit": "TFLOP/s",
                "avg_launch_ms": 2.8971234, "launches_per_step": 1, "traffic": 3.38e9, "kernel_arithmetic": "bf16x3",
                "mfma_per_product": 3, "product_tflops": 294.7, "product_frac_of_f32_mfma_peak": 1.87,
                "issued_gflop_per_launch": 2561.8, "reference_op_gflop_per_launch": 8847.6, "reference_op_tflops": 3054.0,
                "share_of_step": 0.3078, "per_kernel_ms": per, "frac": 0.3536, "traffic_source": "profiles/x.json (k)" * 3,
                "mfma_pipe_busy_pmc": 0.5151, "pmc_launch_ms": 2.93, "clock_ghz": 2.37, "clock_source": "probe " * 60}

    def comp(name):
        return {"value": 57512.123, "unit": "pairs/s", "steps": 20, "warmup": 5, "ms_per_step": 8.9023, "name": name,
                "per_rank_ms_per_step": [8.9023], "dtype": "bf16x3", "max_abs_dlogit_vs_f32_path": 2.2e-5,
                "data": "synthetic " * 9, "metric": "siamese pair-comparisons/sec @1024 pts",
                "config": {"workload": "%s: %s" % (name, "description " * 30), "pairs_per_gpu_per_step": 512,
                           "points": 1024, "backbone_list": [1024, 512, 256], "parallelism": "independent pair shards x1",
                           "rccl_ranks": 1, "precision": prose, "precision_tag": "bf16x3", "launch": "hipgraph",
                           "eager_ms": 8.95, "hipgraph_ms": 8.9, "fill": "kNN groups: every row genuine " * 3},
                "roofline": roof("sa_fused[D=64,c=128/128/128,N=512,S=256,K=48]")}
    full = comp("ssg1024")
    del full["name"]
    full.update(n_gpus=1, higher_is_better=True, scaling="weak", vs_baseline=None,
                roofline=roof("sa_ragged[D=128,c=128/128/256,N=512,S=128,K=64]"))
    full["config"]["fill"] = {"sa1": {"K": 32, "radius": 0.2, "mean_hits": 2.86, "fill": 0.0894},
                              "sa2": {"K": 64, "radius": 0.4, "mean_hits": 7.78, "fill": 0.1216}}
    full["also"] = [comp("companion_%d" % i) for i in range(n_also - 1)] + [{"name": "broken", "error": "E: " + "x" * 900}]
    full["cpu_baseline"] = {"value": 21.1, "unit": "pairs/s", "cores": 32, "threads_used": 32, "host_cores": 256,
                            "kind": "port", "sample": "8 pairs " * 40}
    return full


def test_the_last_stdout_line_is_compact_and_complete(tmp_path):
    import io
    full = _full_record()
    assert len(json.dumps(full)) > 25000                       # (the size that was lost)
    out, err = io.StringIO(), io.StringIO()
    bench.emit(full, out=out, err=err, root=str(tmp_path))      # (never the repository's own bench_full.json: VERDICT r5 item 11)
    lines = out.getvalue().splitlines()
    assert len(lines) == 1                                     # ONE stdout line, and it is the last thing printed
    assert len(lines[0]) < bench.COMPACT_LIMIT and len(lines[0]) < 5500
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"].startswith("ssg1024") and d["config"]["pairs_per_gpu_per_step"] == 512
    assert d["config"]["launch"] == "hipgraph" and d["config"]["precision"] == "bf16x3"
    assert d["config"]["fill"] == {"sa1": 0.089, "sa2": 0.122}
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert d["roofline"]["frac"] == pytest.approx(full["roofline"]["achieved"] / full["roofline"]["peak"], abs=1e-3)
    assert "clock_source" not in d["roofline"] and len(d["roofline"]["per_kernel_ms"]) <= 10
    assert {k for k in ("value", "unit", "cores", "threads_used", "host_cores", "kind", "sample")} <= set(d["cpu_baseline"])
    assert [a["name"] for a in d["also"]] == [a["name"] for a in full["also"]]
    a0 = d["also"][0]
    assert a0["value"] == pytest.approx(57512.1) and a0["ms_per_step"] == 8.902 and a0["frac"] == 0.354
    assert a0["dtype"] == "bf16x3" and "unit" not in a0 and a0["kernel"] == "sa_fused" and "error" in d["also"][-1]
    # the full record survives beside it: one prefixed stderr line + bench_full.json
    blob = err.getvalue()
    assert blob.startswith("bench full record: ") and json.loads(blob[len("bench full record: "):]) == full
    with open(os.path.join(str(tmp_path), bench.FULL_RECORD)) as f:
        assert json.load(f) == full
    assert not os.path.exists(os.path.join(bench.ROOT, bench.FULL_RECORD)) or \
        os.path.getmtime(os.path.join(bench.ROOT, bench.FULL_RECORD)) < os.path.getmtime(os.path.join(str(tmp_path), bench.FULL_RECORD))


def test_an_oversized_record_still_yields_a_parseable_headline(tmp_path):
    import io
    full = _full_record(n_also=60)
    out = io.StringIO()
    bench.emit(full, out=out, err=io.StringIO(), root=str(tmp_path))
    line = out.getvalue().splitlines()[-1]
    assert len(line) < bench.COMPACT_LIMIT
    d = json.loads(line)
    assert d["value"] == full["value"] and d["roofline"]["frac"] and len(d["also"]) == 60


def test_a_record_without_a_roofline_still_yields_the_headline(tmp_path):
    """the overflow path popped from a None roofline (--no-profile / error records) and printed nothing"""
    import io
    full = _full_record(n_also=60)
    full["roofline"] = None
    out = io.StringIO()
    bench.emit(full, out=out, err=io.StringIO(), root=str(tmp_path))
    d = json.loads(out.getvalue().splitlines()[-1])
    assert d["value"] == full["value"] and d["roofline"] is None
    # and a record the compaction cannot digest at all still prints the contract's scalars
    broken = {"metric": "m", "value": 1.5, "unit": "pairs/s", "config": 3, "also": 7}
    out = io.StringIO()
    bench.emit(broken, out=out, err=io.StringIO(), root=str(tmp_path))
    d = json.loads(out.getvalue().splitlines()[-1])
    assert d["metric"] == "m" and d["value"] == 1.5


def test_the_default_workload_is_the_reference_models_config():
    """the headline is the reference's own Point-Transformer config at the metric's 1024 points"""
    import inspect
    src = inspect.getsource(bench.main)
    assert 'default="pt1024"' in src
    for name in ("pt1024_f32", "pt4096_f32", "ssg1024", "ssg1024_full", "pointnet256"):
        assert '"%s"' % name in src
    assert bench.WORKLOADS["pt1024"][3] == [1024, 512, 256] and bench.WORKLOADS["pt1024"][4] == 512


def test_the_sustained_rate_is_informational_and_leaves_peak_and_frac_alone():
    """`roofline.sustained_bf16` quotes the committed probe output (profiles/r*_clock_probe.txt: what a dense bf16 MFMA loop
    sustains with random operands); the contract's `peak` / `frac` are the nominal ones whatever it says"""
    roof = {"bound": "mfma", "kernel_arithmetic": "bf16x3", "achieved": 1150.0, "peak": 2500.0, "frac": 0.46, "unit": "TFLOP/s"}
    bench.add_sustained(roof)
    assert roof["peak"] == 2500.0 and roof["frac"] == 0.46
    s = roof["sustained_bf16"]
    assert s["source"].startswith("profiles/") and s["source"].endswith("_clock_probe.txt")
    assert 2300 < s["constant_operands_tflops"] < 2600          # the nominal peak, reproduced by the probe
    assert 1000 < s["random_operands_tflops"] < s["constant_operands_tflops"]
    assert abs(s["frac_of_random_operands"] - 1150.0 / s["random_operands_tflops"]) < 1e-3
    assert "sustained_bf16" in bench.compact_roofline(roof)
    for other in ({"bound": "hbm", "kernel_arithmetic": "bf16x3", "achieved": 1.0},
                  {"bound": "mfma", "kernel_arithmetic": "f32", "achieved": 1.0}):
        bench.add_sustained(other)
        assert "sustained_bf16" not in other
