"""Small per-sample ops of the model path that are not part of a larger fused launch."""
import torch

from . import _lib as L


def pool_both(x):
    """(B,C,L) -> (B,2C) = [max over L, mean over L]  (ReIDNet.get_pooled_feats, pool_type='both')"""
    L.require_cuda(x)
    x = x.contiguous()
    B, C, Ln = x.shape
    out = torch.empty((B, 2 * C), dtype=torch.float32, device=x.device)
    L.check(L.load().pcr_pool_both_f32(L.ptr(x), L.ptr(out), B, C, Ln, L.stream_ptr()), "pcr_pool_both_f32")
    return out
