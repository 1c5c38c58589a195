"""Reduces a PCR_ATTN_TRACE dump (trace build: PCR_LIB_TAG=trace, -DPCR_SA_TRACE_BUILD) of the wave-autonomous attention
kernels to mean shader clocks per phase.  kv64 (attn_kv_stream64_kernel, one record per cloud round): marks 4 round top,
0 block top, 1 loads + hidden layer + splits done, 2 projection MFMAs done, 3 epilogue + KV MFMAs done (marks 0-3: the
round's LAST block), 5 block loop done, 6 reduction done, 7 fold + stores done.  apply64 (attn_apply_stream64_kernel, one
record per 32-token block): 0 top, 1 loads + splits done, 2 Q MFMAs done, 3 normaliser + message done, 4 LayerNorm 1 done,
5 FFN0 done, 6 FFN1 done, 7 LayerNorm 2 + residual + store done.
usage: trace_attn.py FILE [kv64|apply64] [launch ordinal among the tag's launches, default the one with most records]"""
import sys
import numpy as np
tag = sys.argv[2] if len(sys.argv) > 2 else "kv64"
launches = []
keep = False
for ln in open(sys.argv[1]):
    if ln.startswith("launch"):
        keep = (" %s " % tag) in ln
        if keep:
            launches.append([ln.strip(), []])
        continue
    p = ln.split()
    if keep and p[0] == "wg":
        launches[-1][1].append(np.array([int(x) for x in p[2:]], dtype=np.int64).reshape(-1, 8))
if len(sys.argv) > 3:
    pick = launches[int(sys.argv[3])]
else:
    pick = max(launches, key=lambda l: int(l[0].split(" B ")[1].split()[0]))
print(len(launches), "launches of", tag, ";", pick[0])
rows = np.stack(pick[1])
for wv, nm in ((0, "wave 0"), (1, "wave 5")):
    r = rows[wv::2][:, 1:-1]
    if tag == "kv64":
        nx = rows[wv::2][:, 2:, 4]
        ok = (r[:, :, 4] > 0) & (r[:, :, 7] > r[:, :, 4]) & (nx > r[:, :, 4])
        m = lambda x: float(x[ok].mean())     # noqa: E731
        print(nm, "to last block %.0f | load+hidden+split %.0f | projection %.0f | elu + KV %.0f | reduce %.0f | fold+store %.0f | round to round %.0f"
              % (m(r[:, :, 0] - r[:, :, 4]), m(r[:, :, 1] - r[:, :, 0]), m(r[:, :, 2] - r[:, :, 1]), m(r[:, :, 3] - r[:, :, 2]),
                 m(r[:, :, 6] - r[:, :, 5]), m(r[:, :, 7] - r[:, :, 6]), m(nx - r[:, :, 4])))
    else:
        nx = rows[wv::2][:, 2:, 0]
        ok = (r[:, :, 0] > 0) & (r[:, :, 7] > r[:, :, 0]) & (nx > r[:, :, 0])
        m = lambda x: float(x[ok].mean())     # noqa: E731
        names = ["loads+splits", "Q", "normaliser+message", "LayerNorm 1", "FFN0", "FFN1", "LN 2 + store"]
        print(nm, " | ".join("%s %.0f" % (names[i], m(r[:, :, i + 1] - r[:, :, i])) for i in range(7)), "| block to block %.0f" % m(nx - r[:, :, 0]))
