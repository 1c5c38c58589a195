mkdir -p gpurun_out/r3b
python -m pytest tests/test_gpu_train_variants.py tests/test_gpu_precision.py -q -s -m gpu > gpurun_out/r3b/new.log 2>&1; echo "new rc $?"
grep -E "^\{|passed|failed|Error|error" gpurun_out/r3b/new.log | cut -c1-1500 | tail -30
python -m pytest tests/test_gpu_model.py tests/test_gpu_train_ops.py tests/test_gpu_distributed.py -q -m gpu > gpurun_out/r3b/train.log 2>&1; echo "train rc $?"
tail -5 gpurun_out/r3b/train.log
python bench.py > gpurun_out/r3b/bench_default.json 2> gpurun_out/r3b/bench_default.err; echo "bench rc $?"
tail -3 gpurun_out/r3b/bench_default.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r3b/bench_default.json").read().strip().splitlines()[-1])
print("HEAD", d["dtype"], round(d["value"]), round(d["ms_per_step"],2), d["max_abs_dlogit_vs_f32_path"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d["roofline"]["per_kernel_ms"])
for a in d.get("also", []):
    if "error" in a: print(a["name"], "ERROR", a["error"]); continue
    r=a["roofline"]
    print(a["name"], a.get("dtype"), round(a["value"]), round(a["ms_per_step"],2), a.get("max_abs_dlogit_vs_f32_path"), r["kernel"], round(r["frac"],3))
print("cpu", d.get("cpu_baseline"))
PY
