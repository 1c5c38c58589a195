cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_r03b; mkdir -p $OUT
W=pt128_train
ARGS="--workload $W --steps 3 --warmup 2 --no-cpu-baseline --no-also"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${W}_busy -o t -- python3 bench.py $ARGS > /dev/null 2> $OUT/${W}_busy.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${W}_fetch -o t -- python3 bench.py $ARGS > /dev/null 2> $OUT/${W}_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${W}_write -o t -- python3 bench.py $ARGS > /dev/null 2> $OUT/${W}_write.err
python3 tools/pmc_summary.py $OUT/r03b_${W}_pmc.json 256 $OUT/${W}_busy $OUT/${W}_fetch $OUT/${W}_write > $OUT/${W}_pmc.txt 2>&1
rm -rf $OUT/${W}_busy $OUT/${W}_fetch $OUT/${W}_write
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/prof_r03b/r03b_pt128_train_pmc.json'))
for k,v in d.items():
    if isinstance(v,dict) and ('tstream' in k or 'sa_l1' in k or 'sa_pool' in k):
        print(k[:60], round(v['launch_us'],1), 'us  mfma busy', round(v['mfma_pipe_busy'],3), ' HBM MB', round(v['hbm_bytes_corrected']/1e6,1), ' TB/s', round(v['hbm_bytes_corrected']/1e6/max(v['launch_us'],1e-9),2))
PY
