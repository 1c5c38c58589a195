#!/bin/bash
# same-box A/B: round-5 kernels vs HEAD.  libpcr_hip_r5.so is NOT in the tree; it is made by hand:
#   for f in $(git ls-tree --name-only e010e6d point-cloud-reid_amd/csrc/); do git show e010e6d:$f > /tmp/oldsrc/csrc/$(basename $f); done
#   (pcr_abi_version patched to the binding's number; two stubs returning 0 for pcr_sa_claim_ws_ints / pcr_attn_apply_pool_ok)
#   hipcc --offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -std=c++17 ... -shared -o point-cloud-reid_amd/pcr_amd/lib/libpcr_hip_r5.so
# and removed again afterwards (a tagged library must never ship).
mkdir -p gpurun_out/ab
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], d["config"]["workload"][:7], "step %.3f ms" % d["ms_per_step"], "dominant %.3f ms" % (r.get("avg_launch_ms") or 0), r.get("per_kernel_ms"))'
for rep in 1 2 3; do
  for wl in pt1024 pt4096; do
    PCR_LIB_TAG=r5 PCR_SA_CLAIMS=0 python bench.py --workload $wl --steps 20 --warmup 3 --no-also --no-cpu-baseline 2>/dev/null | python -c "$P" r5
    python bench.py --workload $wl --steps 20 --warmup 3 --no-also --no-cpu-baseline 2>/dev/null | python -c "$P" HEAD
  done
done
