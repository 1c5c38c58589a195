"""GPU: the wave-autonomous attention kernels (attn_kv_stream64 / 32, attn_apply_stream64: whole 32-token blocks at d_model
64 / 32) against the torch-eager oracle over the shapes their template cases and work split depend on -- head counts
(1: full KV tiles, 4: sixteen-channel heads), key sets of 1 / 2 / 3 / 5 blocks (1, 2, 4, 8 waves per cloud, several clouds
per workgroup round), query sets that leave the last workgroup partly filled, batch sizes that are not a multiple of the
clouds per round -- in both arithmetic modes (f32: the kv kernel's f32 form + the tile apply kernel).
Reference: models/pointnet2_utils.py:14-47,90-114; attention.py:192-219."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))

from pcr_amd import engine, testing as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,d,nhead,B,Lq,Sk", [
    ("self", 64, 1, 3, 32, 32), ("self", 64, 2, 7, 96, 96), ("self", 64, 4, 5, 160, 160), ("self", 32, 2, 6, 64, 64),
    ("self", 32, 1, 3, 224, 224), ("cross", 64, 4, 3, 64, 32), ("cross", 64, 2, 9, 32, 160), ("cross", 64, 1, 2, 288, 64),
    ("cross", 32, 2, 5, 96, 256)])
@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_streaming_attention_matches_oracle(kind, d, nhead, B, Lq, Sk, prec):
    import model_oracle as MO
    from mmdet3d.models.attention import corss_attention
    from mmdet3d.models.pointnet2_utils import Self_Attention
    g = torch.Generator().manual_seed(17 * d + Lq + Sk + nhead)
    tt = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
    if kind == "self":
        m = Self_Attention(d, nhead)
        args = (tt(B, d, Lq), tt(B, Lq, 3))
        oracle = MO.self_attention
    else:
        m = corss_attention(d, nhead)
        args = (tt(B, d, Lq), tt(B, Lq, 3), tt(B, d, Sk), tt(B, Sk, 3))
        oracle = MO.cross_attention
    sd = T.seeded_state_dict(T.manifest_of(m), 5)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    with torch.no_grad():
        want = oracle(sd, *args, nhead=nhead)
    with engine.precision(prec), torch.no_grad():
        got = m(*[a.cuda() for a in args]).cpu()
        again = m(*[a.cuda() for a in args]).cpu()
    assert torch.equal(got, again)                                   # fixed reduction orders: run to run identical
    assert float((got - want).abs().max()) < 1e-4, float((got - want).abs().max())
    # one cloud alone gives the bits it gives inside the batch (persistent workgroups, several clouds per round)
    with engine.precision(prec), torch.no_grad():
        one = m(*[a[B - 1:].cuda() for a in args]).cpu()
    assert torch.equal(one[0], got[B - 1])


@pytest.mark.parametrize("nhead,B,L", [(2, 1, 32), (2, 7, 128), (4, 300, 96), (1, 20, 160)])
def test_pooled_apply_equals_the_pooling_of_the_block_output(nhead, B, L):
    """ABI 16: the matching stage's apply launch with `pool_out` (the gallery path, ReIDNet.match_gallery) leaves every
    cloud's per-channel maximum and sum instead of its output tile: the maximum must be the maximum of the values the
    unpooled launch stores (same arithmetic, same bits), the sum their sum up to its order; a cloud alone gives the bits it
    gives inside a batch; the f32 mode (tile kernel) says it cannot pool.  Reference: attention.py:192-219 +
    ReIDNet.py:526-534 (get_pooled_feats 'both')."""
    from mmdet3d.models.attention import corss_attention
    g = torch.Generator().manual_seed(100 * nhead + L)
    m = corss_attention(64, nhead)
    m.load_state_dict(T.seeded_state_dict(T.manifest_of(m), 3))
    m = m.cuda().eval()
    plan = m.plan(torch.device("cuda"))
    feat = torch.randn(B, 64, L, generator=g).cuda()
    xyz = torch.randn(B, L, 3, generator=g).cuda()
    partner = torch.arange(B - 1, -1, -1, dtype=torch.int32).cuda()
    with engine.precision("bf16x3"), torch.no_grad():
        assert plan.pool_ok(L, L)
        kv = plan.kv(feat, xyz)
        o = plan.apply(feat, None, kv, L, kv_index=partner)
        pl = plan.apply(feat, None, kv, L, kv_index=partner, pooled=True)
        assert pl.shape == (B, 2, 64)
        assert torch.equal(pl[:, 0], o.max(dim=2).values)
        ref_sum = o.double().sum(dim=2)
        assert float((pl[:, 1].double() - ref_sum).abs().max()) < 1e-5 * max(1.0, float(ref_sum.abs().max()))
        again = plan.apply(feat, None, kv, L, kv_index=partner, pooled=True)
        assert torch.equal(pl, again)
        one = plan.apply(feat[B - 1:], None, kv, L, kv_index=partner[B - 1:], pooled=True)
        assert torch.equal(one[0], pl[B - 1])
    with engine.precision("f32"):
        assert not plan.pool_ok(L, L)
        kv32 = plan.kv(feat, xyz)
        with pytest.raises(Exception):
            plan.apply(feat, None, kv32, L, kv_index=partner, pooled=True)
