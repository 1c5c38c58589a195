mkdir -p gpurun_out/g2
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "graph_mode or trainer or training" > gpurun_out/g2/test.log 2>&1; echo "tests rc $?"; tail -12 gpurun_out/g2/test.log | cut -c1-300
for p in 256 16; do python bench.py --workload pt128_train --no-cpu-baseline --pairs $p 2>gpurun_out/g2/err_$p.log | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($p, round(d['value']), round(d['ms_per_step'],2), d['config'].get('launch'))"; done
PCR_TRAIN_GRAPH=0 python bench.py --workload pt128_train --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager', round(d['value']), round(d['ms_per_step'],2), d['config'].get('launch'))"
tail -3 gpurun_out/g2/err_256.log
