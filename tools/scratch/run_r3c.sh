mkdir -p gpurun_out/r3c
python -m pytest tests/test_gpu_precision.py tests/test_gpu_train_variants.py -q -s -m gpu > gpurun_out/r3c/new.log 2>&1; echo "new rc $?"
grep -E "^\{|passed|failed|Error" gpurun_out/r3c/new.log | cut -c1-1800 | tail -30
python -m pytest tests/test_gpu_ssg.py tests/test_gpu_model.py tests/test_gpu_config_variants.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -q -m gpu > gpurun_out/r3c/sub.log 2>&1; echo "sub rc $?"
tail -8 gpurun_out/r3c/sub.log
for prec in bf16x3 bf16; do
  for wl in ssg1024 pt1024; do
    PCR_PRECISION=$prec python bench.py --workload $wl --no-also --no-cpu-baseline --detail > gpurun_out/r3c/bench_${wl}_${prec}.json 2> gpurun_out/r3c/bench_${wl}_${prec}.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3c/bench_${wl}_${prec}.json").read().strip().splitlines()[-1])
    print("$wl $prec", round(d["value"]), round(d["ms_per_step"],2), d["max_abs_dlogit_vs_f32_path"], d["roofline"]["per_kernel_ms"])
except Exception as e:
    print("$wl $prec failed", e)
PY
    grep "sa_" gpurun_out/r3c/bench_${wl}_${prec}.err | head -6
  done
done
