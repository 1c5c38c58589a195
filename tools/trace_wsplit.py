"""Reduces a PCR_SA_TRACE dump of sa_wsplit_rag_kernel (trace build: PCR_LIB_TAG=trace, -DPCR_SA_TRACE_BUILD) to mean cycles
per step: marks 0 loop top, 1 layer-2 MFMAs + side work done, 2 X2 stored, 3 past barrier 1, 4 layer 3 + side work done,
5 past barrier 2.  Rows alternate wave 0 / wave 7 of a workgroup."""
import sys
import numpy as np
rows = []
for ln in open(sys.argv[1]):
    if ln.startswith("launch"):
        rows = []          # keep the last launch
        continue
    p = ln.split()
    if p[0] != "wg":
        continue
    v = np.array([int(x) for x in p[6:]], dtype=np.int64).reshape(-1, 8)[:, :6]
    rows.append(v)
rows = np.stack(rows)                      # (2 * wgs, tiles, 6)
ok = (rows[:, :, 5] > 0) & (rows[:, :, 0] > 0)
names = ["step1 mfma+pairs", "x2 store", "barrier 1", "step2 mfma+side", "barrier 2", "loop back"]
for wv, nm in ((0, "wave 0"), (1, "wave 7")):
    r = rows[wv::2][:, 2:-1]               # skip the first tiles (cold) and the last recorded
    k = ok[wv::2][:, 2:-1]
    d = [r[:, :, i + 1] - r[:, :, i] for i in range(5)]
    nxt = rows[wv::2][:, 3:, 0] - rows[wv::2][:, 2:-1, 5]
    print(nm, " ".join("%s %.0f" % (names[i], d[i][k].mean()) for i in range(5)), "loop back %.0f" % nxt[k & ok[wv::2][:, 3:]].mean(),
          "| tile %.0f cycles" % (rows[wv::2][:, 3:, 0] - rows[wv::2][:, 2:-1, 0])[k & ok[wv::2][:, 3:]].mean())
