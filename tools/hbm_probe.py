"""what the box's HBM delivers to simple streaming kernels (context for the HBM-bound rooflines): python tools/hbm_probe.py"""
import torch
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda").normal_()      # 1 GiB
y = torch.empty_like(x)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = x.numel() * 4 / 1e9
ms = t(lambda: y.copy_(x));          print("copy   (read + write): %.3f ms, %.0f GB/s" % (ms, 2 * gb / ms * 1e3))
ms = t(lambda: x.sum());             print("sum    (read only)   : %.3f ms, %.0f GB/s" % (ms, gb / ms * 1e3))
ms = t(lambda: y.fill_(1.0));        print("fill   (write only)  : %.3f ms, %.0f GB/s" % (ms, gb / ms * 1e3))
ms = t(lambda: torch.add(x, y, out=y)); print("add    (2 reads + write): %.3f ms, %.0f GB/s" % (ms, 3 * gb / ms * 1e3))
