mkdir -p gpurun_out/r3d
python -m pytest tests/test_gpu_train_variants.py tests/test_gpu_model.py -q -s -m gpu > gpurun_out/r3d/new.log 2>&1; echo "new rc $?"
grep -E "passed|failed|Error" gpurun_out/r3d/new.log | cut -c1-400 | tail -12
grep -o '{"losses".*"dlogits".*}' gpurun_out/r3d/new.log | cut -c1-900
export PCR_LIB_TAG=tune
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --workload $WL --no-also --no-cpu-baseline --detail --steps 8 > gpurun_out/r3d/$name.json 2> gpurun_out/r3d/$name.err
  echo "== $WL $name $(python -c "import json;d=json.loads(open('gpurun_out/r3d/$name.json').read().strip().splitlines()[-1]);print(round(d['value']),round(d['ms_per_step'],2))" 2>/dev/null)"
  grep "^sa_\(fused\|ragged\)" gpurun_out/r3d/$name.err | awk '{print "   ",$1,$2,$3}'
}
WL=pt1024
for cpw in 1 2 3 4 6; do run pt_cpw$cpw PCR_SA_CPW=$cpw; done
for dbg in 1 2 4 3; do run pt_dbg$dbg PCR_SA_DBG=$dbg; done
run pt_bf16_dbg2 PCR_PRECISION=bf16 PCR_SA_DBG=2
WL=ssg1024
for dbg in 0 1 2 4 8 6; do run ssg_dbg$dbg PCR_SA_DBG=$dbg; done
