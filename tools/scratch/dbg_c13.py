import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "point-cloud-reid_amd"))
import torch
import bench
from pcr_amd import engine
model, sd = bench.build_pt_model([1024, 512, 256])
model = model.cuda().eval()
import gc
x = torch.randn(4, 1024, 3).cuda()
with torch.no_grad():
    model.forward_inference(x)
plans = [o for o in gc.get_objects() if isinstance(o, engine.AttnPlan)]
for pl in plans:
    print("plan d", pl.d, "c1", pl.c1, "c2", pl.c2, "cout", pl.cout, "cfinal", pl.cfinal, "qpos", pl.q_pos, "res", pl.residual, "nhead", pl.nhead, "xpad" , "wmlp0_bf_xpad" in pl.t, "wkv_bf", "wkv_bf" in pl.t)
