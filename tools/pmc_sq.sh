#!/bin/bash
# LDS / VALU / MFMA activity counters per kernel for one workload (diagnostics)
set -u
W=${1:-pt1024}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$W; mkdir -p $OUT
ARGS="--workload $W --steps 3 --warmup 2 --no-cpu-baseline --no-also"
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o t -- python3 bench.py $ARGS > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/b -o t -- python3 bench.py $ARGS > /dev/null 2> $OUT/b.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU2 --output-format csv -d $OUT/c -o t -- python3 bench.py $ARGS > /dev/null 2> $OUT/c.err
python3 - $W <<'PY' > $OUT/summary.txt
import csv, glob, collections, os, sys, json
W = sys.argv[1]
res = {}
for d in ("a", "b", "c"):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/pmc_%s/%s/**/*counter_collection.csv" % (W, d), recursive=True):
        for r in csv.DictReader(open(f)):
            per[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, c in per.items():
        key = "SQ_BUSY_CYCLES" if "SQ_BUSY_CYCLES" in c else "SQ_INSTS_VALU"
        vals = sorted(c[key])
        # the full-batch launches are the largest of the process: the dispatch position (rank in dispatch order) of the max
        pos = max(range(len(vals)), key=lambda i: vals[i][1])
        for n, v in c.items():
            v = sorted(v)
            res.setdefault(k, {})[n] = v[pos][1] if pos < len(v) else None
json.dump(res, open("gpurun_out/pmc_%s/raw.json" % W, "w"))
for k, m in sorted(res.items(), key=lambda kv: -(kv[1].get("GRBM_GUI_ACTIVE") or 0)):
    if not any(t in k for t in ("sa_", "attn", "knn")): continue
    gui = m.get("GRBM_GUI_ACTIVE") or 1.0
    xcd = gui / 8
    f = lambda n, units: (m.get(n) or 0.0) / (units * xcd)
    print("%-92s %7.1f us | mfma %.2f valu %.2f lds_idx %.2f lds_inst %.2f wait_lds %.2f bankconf %.3f | per-wave: wait_any %.2f | coexec %.3f valu2 %.3f | insts valu %.3g mfma %.3g lds %.3g salu %.3g" % (
        k.replace("(anonymous namespace)::", "")[:92], xcd / 2.4e3, f("SQ_VALU_MFMA_BUSY_CYCLES", 1024), f("SQ_ACTIVE_INST_VALU", 1024) * 4, f("SQ_LDS_IDX_ACTIVE", 256),
        f("SQ_ACTIVE_INST_LDS", 1024) * 4, f("SQ_WAIT_INST_LDS", 1024) * 4, f("SQ_LDS_BANK_CONFLICT", 256),
        (m.get("SQ_WAIT_ANY") or 0) / (m.get("SQ_WAVE_CYCLES") or 1), f("SQ_VALU_MFMA_COEXEC_CYCLES", 1024), f("SQ_ACTIVE_INST_VALU2", 1024) * 4, m.get("SQ_INSTS_VALU") or 0, m.get("SQ_INSTS_MFMA") or 0, m.get("SQ_INSTS_LDS") or 0, m.get("SQ_INSTS_SALU") or 0))
PY
sed -i 's/^/ /' $OUT/summary.txt
rm -rf $OUT/a $OUT/b $OUT/c
