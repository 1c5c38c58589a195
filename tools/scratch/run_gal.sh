#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/gal
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gal/stats -o t -- python3 bench.py --workload gallery128 --steps 5 --warmup 2 --no-cpu-baseline --no-also > gpurun_out/gal/bench.json 2> gpurun_out/gal/err.log
head -25 gpurun_out/gal/stats/t_kernel_stats.csv | cut -c1-160
tail -1 gpurun_out/gal/bench.json | cut -c1-400
rm -f gpurun_out/gal/stats/t_kernel_trace.csv
