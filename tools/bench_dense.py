"""times one wide per-point layer (pcr_dense_gn_prec_f32 / pcr_dense_prec_f32) at a PointNet / LinearRes shape (diagnostics;
PCR_LIB_TAG=tune + PCR_DPC_DBG ablate the two-role kernel, PCR_DENSE_NO_PC=1 takes the one-role kernel):
python tools/bench_dense.py [cin] [cout] [L] [B] [gn]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "point-cloud-reid_amd")]
import torch
from pcr_amd import engine as E, rows
cin = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 512
Ln = int(sys.argv[3]) if len(sys.argv) > 3 else 256
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
use_gn = (sys.argv[5] != "0") if len(sys.argv) > 5 else True
x = torch.randn(B, cin, Ln, device="cuda")
w = torch.randn(cout, cin) / cin ** 0.5
wp = E.pack_weight_dual(w, "cuda")
gn = torch.nn.GroupNorm(cout // 8, cout).cuda()
res = torch.randn(B, cout, Ln, device="cuda")
sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
def run():
    return rows.dense_gn(x, wp, cout, gn, res=res, relu=True) if use_gn else E.dense(x, wp, cout, sc, sh, act=1)
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("dbg=%s nopc=%s cin=%d cout=%d L=%d B=%d gn=%d: %.3f ms, %.0f TFLOP/s issued (x3)" % (
    os.environ.get("PCR_DPC_DBG", "0"), os.environ.get("PCR_DENSE_NO_PC", "0"), cin, cout, Ln, B, use_gn, ms,
    3 * 2.0 * B * Ln * cin * cout / ms / 1e9))
