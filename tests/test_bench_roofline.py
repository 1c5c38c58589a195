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
    PCR_BENCH_GRAPH=0, it must hand back the callable it was given and say why"""
    import bench
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
