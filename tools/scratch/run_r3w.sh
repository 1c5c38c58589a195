mkdir -p gpurun_out/r3w
python -m pytest tests/test_gpu_point_ops.py tests/test_gpu_ssg.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r3w/test.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r3w/test.log | cut -c1-200
bash tools/collect_profiles.sh r03b "ssg1024" > gpurun_out/r3w/collect.log 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/prof_r03b/r03b_ssg1024_pmc.json'))
for k,v in d.items():
    if 'ball' in k or 'rag_rows' in k or 'fps' in k: print(k[:70], v)
PY
tail -c 1500 gpurun_out/prof_r03b/ssg1024_bench.json | head -c 600
