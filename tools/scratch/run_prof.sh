bash tools/collect_profiles.sh r03b "ssg1024 pt1024 pt128_train" > gpurun_out/prof_r03b_collect.log 2>&1
python bench.py > gpurun_out/prof_r03b/r03b_default_bench.json 2> gpurun_out/prof_r03b/default_bench.err; echo "default bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/prof_r03b/r03b_default_bench.json').read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],2), d['dtype'], d['roofline'].get('frac'))
for k,v in d.get('also',{}).items():
    print(k, round(v.get('value',0)), round(v.get('ms_per_step',0),2))
PY
for w in pt128 gallery128 pointnet256 dgcnn256; do python bench.py --workload $w --no-also --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['value']), round(d['ms_per_step'],2))"; done
