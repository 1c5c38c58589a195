"""times one train-dense forward launch at a grouped-MLP shape (diagnostics): python tools/bench_tdense_fwd.py [c] [L] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
from pcr_amd import train_ops as TO, _lib
if os.environ.get('PCR_STREAM_MIN'): _lib.load().pcr_set_stream_min_blocks(int(os.environ['PCR_STREAM_MIN']))
c = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Ln = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
x = torch.randn(B, c, Ln, device="cuda")
W = torch.randn(c, c, device="cuda") / c ** 0.5
bias = torch.randn(c, device="cuda")
isc, ish = torch.rand(c, device="cuda") + 0.5, torch.randn(c, device="cuda") * 0.1
wp = TO.pack_dev(W)
def run():
    return TO.tdense_fwd(x, wp, c, isc=isc, ish=ish, in_relu=True, bias=bias, want_stats=True)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("c=%d L=%d B=%d: %.3f ms per call, %.1f GB/s algorithmic, %.1f TFLOP/s" % (
    c, Ln, B, ms, 4.0 * B * Ln * 2 * c / ms / 1e6, 2.0 * B * Ln * c * c / ms / 1e9))
