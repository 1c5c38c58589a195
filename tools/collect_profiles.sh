#!/bin/bash
# rocprofv3 summaries of one round (run on the GPU box through gpurun): kernel stats + the three PMC passes per workload.
#   bash tools/collect_profiles.sh r02a "ssg1024 pt1024 pt128_train"
# Counters are collected in their own runs (--pmc with --kernel-trace only), as the pool requires.
set -u
TAG=$1; WLS=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
for W in $WLS; do
  ARGS="--workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-also"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${W}_stats -o t -- python3 bench.py $ARGS > $OUT/${W}_bench.json 2> $OUT/${W}_stats.err
  cp $OUT/${W}_stats/t_kernel_stats.csv $OUT/${TAG}_${W}_kernel_stats.csv
  if [ "$W" != "pt128_train" ]; then
    rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${W}_busy -o t -- python3 bench.py $ARGS > /dev/null 2> $OUT/${W}_busy.err
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${W}_fetch -o t -- python3 bench.py $ARGS > /dev/null 2> $OUT/${W}_fetch.err
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${W}_write -o t -- python3 bench.py $ARGS > /dev/null 2> $OUT/${W}_write.err
    PAIRS=$(python3 -c "import json,sys; print(json.loads(open('$OUT/${W}_bench.json').read().strip().splitlines()[-1])['config']['pairs_per_gpu_per_step'])")
    python3 tools/pmc_summary.py $OUT/${TAG}_${W}_pmc.json $PAIRS $OUT/${W}_busy $OUT/${W}_fetch $OUT/${W}_write ${PCR_PRECISION:-bf16x3} > $OUT/${W}_pmc.txt 2>&1
    rm -rf $OUT/${W}_busy $OUT/${W}_fetch $OUT/${W}_write
  fi
  rm -rf $OUT/${W}_stats
done
ls -la $OUT
