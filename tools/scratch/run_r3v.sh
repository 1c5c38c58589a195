mkdir -p gpurun_out/r3v
python -m pytest tests -q -m gpu > gpurun_out/r3v/all.log 2>&1; echo "all rc $?"
tail -4 gpurun_out/r3v/all.log
python bench.py > gpurun_out/r3v/bench_default.json 2> gpurun_out/r3v/bench_default.err; echo "bench rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r3v/bench_default.json").read().strip().splitlines()[-1])
print("HEAD", d["dtype"], round(d["value"]), round(d["ms_per_step"],2), d["max_abs_dlogit_vs_f32_path"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
for a in d.get("also", []):
    if "error" in a: print(a["name"], "ERROR", a["error"]); continue
    r=a["roofline"]
    print(a["name"], a.get("dtype"), round(a["value"]), round(a["ms_per_step"],2), a.get("max_abs_dlogit_vs_f32_path"), r["kernel"], round(r["frac"],3))
print("cpu", d.get("cpu_baseline",{}).get("value"))
PY
bash tools/collect_profiles.sh r03e "ssg1024 pt1024 pt128_train" > gpurun_out/r3v/prof.log 2>&1
tail -5 gpurun_out/r3v/prof.log
cat gpurun_out/prof_r03e/ssg1024_pmc.txt | head -12
cat gpurun_out/prof_r03e/pt1024_pmc.txt | head -14
