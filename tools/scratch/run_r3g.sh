for args in "--steps 20" "--steps 200" "--steps 20 --warmup 50"; do
  for prec in bf16x3 f32; do
    PCR_PRECISION=$prec python bench.py --workload pt128 --no-also --no-cpu-baseline $args 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$prec $args', round(d['value']), round(d['ms_per_step'],3))"
  done
done
