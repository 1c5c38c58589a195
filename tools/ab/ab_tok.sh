#!/bin/bash
# same box: round-5 kernels / HEAD / HEAD tuning build with the MFMA token on and off / claims off
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("%-14s" % sys.argv[1], d["config"]["workload"][:7], "step %.3f ms" % d["ms_per_step"], "dominant %.3f ms" % (r.get("avg_launch_ms") or 0), "sa_fused", r.get("per_kernel_ms",{}).get("sa_fused"))'
B="python bench.py --workload pt1024 --steps 20 --warmup 3 --no-also --no-cpu-baseline"
for rep in 1 2 3; do
  PCR_LIB_TAG=r5 PCR_SA_CLAIMS=0 $B 2>/dev/null | python -c "$P" r5
  $B 2>/dev/null | python -c "$P" HEAD
  PCR_LIB_TAG=tune $B 2>/dev/null | python -c "$P" tune
  PCR_LIB_TAG=tune PCR_SA_DBG=1024 $B 2>/dev/null | python -c "$P" tune_tok_off
  PCR_LIB_TAG=tune PCR_SA_DBG=4096 $B 2>/dev/null | python -c "$P" tune_claim_off
  PCR_LIB_TAG=tune PCR_SA_DBG=5120 $B 2>/dev/null | python -c "$P" tune_both_off
done
