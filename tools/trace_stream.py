"""Reduces a PCR_SA_TRACE dump of the wave-autonomous SA kernels (trace build: PCR_LIB_TAG=trace, -DPCR_SA_TRACE_BUILD) to
mean shader clocks per phase of a 32-row block.  Marks: 0 block top, 6 layer 1 done (index / table gathers included),
7 layer 2 done, 1 layer 3 done, 2 maxima stored (K-row form: group maxima; ragged form: pair maxima), 3 (ragged form)
reduction over pairs done; 4 = top of the block's item, 5 (ragged) its row map built.  Rows alternate wave 0 / wave 5.
usage: trace_stream.py FILE [stream|krow] [launch ordinal among the tag's launches, default last]"""
import sys
import numpy as np
tag = sys.argv[2] if len(sys.argv) > 2 else "stream"
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
launches, keep = [], False
for ln in open(sys.argv[1]):
    if ln.startswith("launch"):
        keep = (" %s " % tag) in ln
        if keep:
            launches.append([])
        continue
    p = ln.split()
    if not keep or p[0] != "wg":
        continue
    launches[-1].append(np.array([int(x) for x in p[6:]], dtype=np.int64).reshape(-1, 8))
print("%d launches of %s" % (len(launches), tag))
rows = np.stack(launches[which])                      # (2 * wgs, blocks, 8)
for wv, nm in ((0, "wave 0"), (1, "wave 5")):
    r = rows[wv::2][:, 2:-1]
    nx = rows[wv::2][:, 3:, 0]
    ok = (r[:, :, 0] > 0) & (r[:, :, 2] > r[:, :, 0]) & (nx > r[:, :, 0])
    m = lambda x: float(x[ok].mean())     # noqa: E731
    out = ["layer 1 (+ gathers) %.0f" % m(r[:, :, 6] - r[:, :, 0]), "layer 2 %.0f" % m(r[:, :, 7] - r[:, :, 6]),
           "layer 3 %.0f" % m(r[:, :, 1] - r[:, :, 7]), "maxima %.0f" % m(r[:, :, 2] - r[:, :, 1])]
    if tag == "stream":
        out.append("reduce %.0f" % m(r[:, :, 3] - r[:, :, 2]))
    elif (r[:, :, 3][ok] > 0).all():      # K-row form, round 6: mark 3 = MFMA token taken (inside 'layer 2')
        out.append("of layer 2: wait for the token %.0f" % m(r[:, :, 3] - r[:, :, 6]))
    out.append("block to block %.0f" % m(nx - r[:, :, 0]))
    if tag == "krow" and (r[:, :, 5][ok] > 0).all():      # round 6: mark 5 = s_memrealtime (100 MHz) at the block top
        full = rows[wv::2]
        dc = (full[:, -1, 0] - full[:, 0, 0]).astype(np.float64)
        dr = (full[:, -1, 5] - full[:, 0, 5]).astype(np.float64)
        good = (dr > 0) & (dc > 0)
        out.append("shader clock %.0f MHz" % (100.0 * float(np.median(dc[good] / dr[good]))))
    print(nm, " | ".join(out))
