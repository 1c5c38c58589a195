#!/bin/bash
# same-box A/B of a tagged library against the product one, separate processes alternating (clock drift between processes on
# these boxes is +-1.5 %: read the per-kernel columns, three alternations).  usage: ab_tag.sh TAG [workload ...]
# The tagged library is built by hand (PCR_LIB_TAG=TAG PCR_EXTRA_HIPCC_FLAGS=-D... python -m pcr_amd.build) and never ships.
TAG=$1; shift
WLS=${@:-pt1024}
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("%-6s" % sys.argv[1], d["config"]["workload"][:7], "step %.3f ms" % d["ms_per_step"], "dominant %.3f ms" % (r.get("avg_launch_ms") or 0), "dlogit %.2e" % (d.get("max_abs_dlogit_vs_f32_path") or 0), r.get("per_kernel_ms"))'
for rep in 1 2 3; do
  for wl in $WLS; do
    PCR_LIB_TAG=$TAG python bench.py --workload $wl --steps 20 --warmup 3 --no-also --no-cpu-baseline 2>/dev/null | python -c "$P" $TAG
    python bench.py --workload $wl --steps 20 --warmup 3 --no-also --no-cpu-baseline 2>/dev/null | python -c "$P" HEAD
  done
done
