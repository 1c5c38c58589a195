mkdir -p gpurun_out/r3a
python -m pytest tests/test_gpu_precision.py -q -x -s -m gpu > gpurun_out/r3a/prec.log 2>&1; echo "prec rc $?"
tail -30 gpurun_out/r3a/prec.log
python -m pytest tests -q -m gpu > gpurun_out/r3a/all.log 2>&1; echo "all rc $?"
tail -15 gpurun_out/r3a/all.log
for prec in f32 bf16x3 bf16; do
  for wl in ssg1024 pt1024; do
    PCR_PRECISION=$prec python bench.py --workload $wl --no-also --no-cpu-baseline --detail > gpurun_out/r3a/bench_${wl}_${prec}.json 2> gpurun_out/r3a/bench_${wl}_${prec}.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r3a/bench_${wl}_${prec}.json").read().strip().splitlines()[-1])
    print("$wl $prec", round(d["value"]), round(d["ms_per_step"],2), d["roofline"]["per_kernel_ms"])
except Exception as e:
    print("$wl $prec failed", e)
PY
  done
done
