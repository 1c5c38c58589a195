"""GPU: claimed work items of the wave-autonomous K-row SA kernel (pcr_sa_params.claim_ws, ABI 16).  On clouds of >= 1024
points the waves of an XCD slot take their items from a shared counter instead of a fixed stride, so that they stay on
consecutive items and one or two clouds' tables live in the L2 instead of three or four (profiles/r06f_pt4096_pmc.json:
2.9 GB per launch instead of 7.2 GB).  Which wave evaluates an item cannot change its arithmetic: the outputs must be
bit-identical, for every shape the query says yes to, for partial last items and for batches that do not fill the chip.
Reference semantics of the layer: models/pointnet2_utils.py:242-288, 333-360."""
import pytest
import torch
import torch.nn as nn

from pcr_amd import _lib as L
from pcr_amd import engine

pytestmark = pytest.mark.gpu


def _layer(D, c, seed):
    g = torch.Generator().manual_seed(seed)
    cin = 3 + 2 * D
    convs = [nn.Conv2d(a, b, 1) for a, b in ((cin, c), (c, c), (c, c))]
    bns = [nn.BatchNorm2d(c) for _ in range(3)]
    for bn in bns:
        bn.running_mean.copy_(torch.randn(c, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
        bn.weight.data.copy_(torch.rand(c, generator=g) + 0.5)
        bn.bias.data.copy_(torch.randn(c, generator=g) * 0.1)
        bn.eval()
    return engine.SaPlan(convs, bns, torch.device("cuda"), 0)


def test_the_query_is_shape_only():
    lib = L.load()
    bf = engine.PRECISIONS["bf16x3"]
    assert lib.pcr_sa_claim_ws_ints(64, 64, 64, 48, 4096, bf) == 8192
    assert lib.pcr_sa_claim_ws_ints(32, 32, 32, 32, 2048, bf) == 8192
    assert lib.pcr_sa_claim_ws_ints(64, 64, 64, 48, 1024, bf) == 8192
    assert lib.pcr_sa_claim_ws_ints(64, 64, 64, 48, 512, bf) == 0           # small clouds: short items, the claim's latency shows
    assert lib.pcr_sa_claim_ws_ints(128, 128, 128, 48, 4096, bf) == 0       # 128 channels: measured, no gain
    assert lib.pcr_sa_claim_ws_ints(64, 64, 64, 48, 4096, 0) == 0           # f32: the tile kernel


@pytest.mark.parametrize("D,c,K,B,N,S", [(32, 64, 48, 9, 1024, 512), (32, 64, 48, 5, 2048, 1000), (0, 32, 32, 3, 4096, 4096), (32, 64, 16, 2, 2048, 2048),
                                         (0, 32, 48, 70, 2048, 301)])
def test_claimed_items_give_the_bits_of_dealt_items(D, c, K, B, N, S, monkeypatch):
    g = torch.Generator().manual_seed(N + S + K)
    xyz = torch.randn(B, N, 3, generator=g).cuda()
    feat = torch.randn(B, D, N, generator=g).cuda() if D else None
    plan = _layer(D, c, 7)
    with engine.precision("bf16x3"), torch.no_grad():
        idx = engine.knn_prefix(xyz, S, K)
        monkeypatch.setattr(engine, "SA_CLAIMS", True)
        a = plan.run(xyz, feat, idx)
        a2 = plan.run(xyz, feat, idx)
        monkeypatch.setattr(engine, "SA_CLAIMS", False)
        b = plan.run(xyz, feat, idx)
    assert torch.isfinite(a).all() and torch.equal(a, a2) and torch.equal(a, b)
