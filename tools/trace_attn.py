"""Reduces a PCR_ATTN_TRACE dump of attn_kv_stream64_kernel (trace build) to mean shader clocks per phase of a cloud round:
marks 4 round top, 0 block top, 1 loads + hidden layer + splits done, 2 projection MFMAs done, 3 epilogue + KV MFMAs done,
5 block loop done, 6 wave-order reduction done, 7 fold + stores done.  usage: trace_attn.py FILE [launch ordinal]"""
import sys
import numpy as np
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
launches = []
for ln in open(sys.argv[1]):
    if ln.startswith("launch"):
        launches.append([ln.strip(), []])
        continue
    p = ln.split()
    if p[0] == "wg":
        launches[-1][1].append(np.array([int(x) for x in p[2:]], dtype=np.int64).reshape(-1, 8))
print(len(launches), "launches;", launches[which][0])
rows = np.stack(launches[which][1])
for wv, nm in ((0, "wave 0"), (1, "wave 5")):
    r = rows[wv::2][:, 1:-1]
    nx = rows[wv::2][:, 2:, 4]
    ok = (r[:, :, 4] > 0) & (r[:, :, 7] > r[:, :, 4]) & (nx > r[:, :, 4])
    m = lambda x: float(x[ok].mean())     # noqa: E731
    print(nm, "to block %.0f | load+hidden+split %.0f | projection %.0f | elu + KV %.0f | (loop end %.0f) | reduce %.0f | fold+store %.0f | round to round %.0f"
          % (m(r[:, :, 0] - r[:, :, 4]), m(r[:, :, 1] - r[:, :, 0]), m(r[:, :, 2] - r[:, :, 1]), m(r[:, :, 3] - r[:, :, 2]),
             m(r[:, :, 5] - r[:, :, 3]), m(r[:, :, 6] - r[:, :, 5]), m(r[:, :, 7] - r[:, :, 6]), m(nx - r[:, :, 4])))
