#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r3t
python -m pytest tests/test_gpu_model.py tests/test_gpu_gallery.py tests/test_gpu_ssg.py -x -q -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/r3t/bench_default.json 2> gpurun_out/r3t/bench_default.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r3t/bench_default.json").read().strip().splitlines()[-1])
print("HEAD", d["dtype"], round(d["value"]), round(d["ms_per_step"],2))
for a in d.get("also", []):
    if "error" in a: print(a["name"], "ERROR", a["error"]); continue
    print(a["name"], a.get("dtype"), round(a["value"]), round(a["ms_per_step"],2))
PY
