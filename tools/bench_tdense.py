"""times one train-dense backward launch at a grouped-MLP shape (diagnostics; PCR_TD_DBG ablates phases):
python tools/bench_tdense.py [c] [L] [B]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
from pcr_amd import train_ops as TO, _lib
if os.environ.get('PCR_STREAM_MIN'): _lib.load().pcr_set_stream_min_blocks(int(os.environ['PCR_STREAM_MIN']))
c = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Ln = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
g = torch.randn(B, c, Ln, device="cuda"); y = torch.randn(B, c, Ln, device="cuda"); x = torch.randn(B, c, Ln, device="cuda")
W = torch.randn(c, c, device="cuda") / c ** 0.5
k = dict(ka=torch.rand(c, device="cuda"), kb=torch.rand(c, device="cuda") * 0.01, kc=torch.rand(c, device="cuda") * 0.01)
isc, ish, iinv = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.1, torch.rand(c, device="cuda") + 0.5
wpT = TO.pack_dev(W, transpose=True)
wpT_bf = TO.pack_bf_T(W) if (c == 128 and TO.TRAIN_PRECISION != "f32") else None      # (the 128 x 128 launch on the bf16 core)
def run():
    return TO.tdense_bwd(g, x, c, dy_mode=1, y=y, k=k, isc=isc, ish=ish, iinv=iinv, in_relu=True, wpT=wpT, want_dstats=True,
                         wpT_bf=wpT_bf)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("dbg=%s c=%d L=%d B=%d: %.3f ms per call (incl. the reduce launch), %.1f GB/s algorithmic" % (
    os.environ.get("PCR_TD_DBG", "0"), c, Ln, B, ms, 4.0 * B * Ln * 4 * c / ms / 1e6))
