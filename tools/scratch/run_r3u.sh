mkdir -p gpurun_out/r3u
python -m pytest tests/test_gpu_attn_split.py -x -q -m gpu > gpurun_out/r3u/test.log 2>&1; echo "tests rc $?"; tail -4 gpurun_out/r3u/test.log
for ns in 1 2 3 4; do
PCR_KV_SPLITS=$ns python bench.py --workload pt1024 --no-also --no-cpu-baseline --detail 2> gpurun_out/r3u/pt1024_$ns.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($ns, round(d['value']), round(d['ms_per_step'],2), d['roofline'].get('per_kernel_ms'))"
grep -E "attn_kv" gpurun_out/r3u/pt1024_$ns.err | head -3
done
