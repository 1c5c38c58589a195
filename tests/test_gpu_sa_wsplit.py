"""GPU: the cout-split grouped-SA kernel with register-resident weights (sa_wsplit_rag_kernel: 128 / 128 / 256, SSG's
second set-abstraction layer, point_sa_module.py:166-216 over ball-query groups) against plain torch fp32 over what its
tile plan and software pipeline depend on: hit counts (1 .. K, all-ones, all-full), K = 16 / 32 / 64, with and without a
feature table, centre counts that leave partial tiles, a single cloud, more clouds than XCDs, both output layouts -- and
ragged evaluation must equal the K-row evaluation of the same launch bit for bit (one kernel, one arithmetic)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _layer(D, g):
    convs = [nn.Conv2d(a, b, 1) for a, b in ((3 + D, 128), (128, 128), (128, 256))]
    bns = [nn.BatchNorm2d(c) for c in (128, 128, 256)]
    for bn in bns:
        bn.running_mean.copy_(torch.randn(bn.num_features, generator=g) * 0.1)
        bn.running_var.copy_(torch.rand(bn.num_features, generator=g) + 0.5)
        bn.weight.data.copy_(torch.rand(bn.num_features, generator=g) + 0.5)
        bn.bias.data.copy_(torch.randn(bn.num_features, generator=g) * 0.1)
        bn.eval()
    return convs, bns


def _groups(B, N, S, K, kind, g):
    """ball-query shaped groups: cnt genuine entries, the rest repeat entry 0 (ball_query_cuda.cu:43-47)"""
    idx = torch.randint(0, N, (B, S, K), generator=g, dtype=torch.int32)
    if kind == "ones":
        cnt = torch.ones(B, S, dtype=torch.int32)
    elif kind == "full":
        cnt = torch.full((B, S), K, dtype=torch.int32)
    else:
        cnt = torch.randint(1, K + 1, (B, S), generator=g, dtype=torch.int32)
        cnt[:, ::7] = 1
        cnt[:, 3::11] = K
    k = torch.arange(K).view(1, 1, K)
    idx = torch.where(k < cnt.unsqueeze(-1), idx, idx[:, :, :1].expand(-1, -1, K))
    return idx.contiguous(), cnt.contiguous()


def _torch_ref(xyz, feat, centre, idx, convs, bns):
    B, S, K = idx.shape
    il = idx.long()
    with torch.no_grad():
        nb = torch.gather(xyz, 1, il.reshape(B, S * K, 1).expand(-1, -1, 3)).view(B, S, K, 3)
        rows = [nb - centre.unsqueeze(2)]
        if feat is not None:
            D = feat.shape[1]
            rows.append(torch.gather(feat.permute(0, 2, 1), 1, il.reshape(B, S * K, 1).expand(-1, -1, D)).view(B, S, K, D))
        x = torch.cat(rows, dim=-1).permute(0, 3, 1, 2)
        for c, b in zip(convs, bns):
            x = torch.relu(b(c(x)))
        return x.max(dim=3)[0]


@pytest.mark.parametrize("B,S,K,D,kind", [(5, 70, 64, 128, "rand"), (1, 33, 64, 128, "rand"), (11, 128, 32, 128, "rand"),
                                          (3, 50, 16, 0, "rand"), (4, 64, 64, 128, "ones"), (2, 40, 64, 128, "full"),
                                          (9, 17, 48, 64, "rand")])
def test_cout_split_kernel_against_torch_and_k_row(B, S, K, D, kind):
    from pcr_amd import engine
    g = torch.Generator().manual_seed(100 * B + S + K)
    N = 300
    xyz = torch.randn(B, N, 3, generator=g)
    feat = torch.randn(B, D, N, generator=g) if D else None
    convs, bns = _layer(D, g)
    idx, cnt = _groups(B, N, S, K, kind, g)
    cidx = torch.randint(0, N, (B, S), generator=g, dtype=torch.int32)
    centre = torch.gather(xyz, 1, cidx.long().unsqueeze(-1).expand(-1, -1, 3))
    want = _torch_ref(xyz, feat, centre, idx, convs, bns)
    plan = engine.SaPlan(convs, bns, torch.device("cuda"), 1)
    dev = lambda t: None if t is None else t.cuda()     # noqa: E731
    scale = float(want.abs().max())
    for prec, tol in (("bf16x3", 2e-5), ("bf16", 3e-2)):
        with engine.precision(prec), torch.no_grad():
            rag = plan.run(dev(xyz), dev(feat), dev(idx), centre_idx=dev(cidx), cnt=dev(cnt))
            rag_pm = plan.run(dev(xyz), dev(feat), dev(idx), centre_idx=dev(cidx), cnt=dev(cnt), out_point_major=True)
            krow = plan.run(dev(xyz), dev(feat), dev(idx), centre_idx=dev(cidx))
        assert float((rag.cpu() - want).abs().max()) / scale < tol, prec
        assert torch.equal(rag, krow), prec                           # ragged == K-row, bit for bit
        assert torch.equal(rag_pm.contiguous(), rag), prec            # both output layouts
    # the launch really is the cout-split kernel's shape (the query the engine sizes its workspace by)
    from pcr_amd import _lib as L
    assert L.load().pcr_sa_krow_uses_tiles(128, 128, 256, K, 1) == 1
    assert L.load().pcr_sa_krow_uses_tiles(128, 128, 256, K, 0) == 0 and L.load().pcr_sa_krow_uses_tiles(64, 64, 128, K, 1) == 0


def test_cout_split_kernel_is_deterministic_and_batch_independent():
    """a cloud evaluated alone gives the bits it gives inside a batch, and twice the same launch gives the same bits"""
    from pcr_amd import engine
    g = torch.Generator().manual_seed(77)
    B, N, S, K, D = 6, 256, 96, 64, 128
    xyz = torch.randn(B, N, 3, generator=g)
    feat = torch.randn(B, D, N, generator=g)
    convs, bns = _layer(D, g)
    idx, cnt = _groups(B, N, S, K, "rand", g)
    plan = engine.SaPlan(convs, bns, torch.device("cuda"), 1)
    with torch.no_grad():
        a = plan.run(xyz.cuda(), feat.cuda(), idx.cuda(), cnt=cnt.cuda())
        b = plan.run(xyz.cuda(), feat.cuda(), idx.cuda(), cnt=cnt.cuda())
        one = plan.run(xyz[2:3].cuda(), feat[2:3].contiguous().cuda(), idx[2:3].cuda(), cnt=cnt[2:3].cuda())
    assert torch.equal(a, b) and torch.equal(a[2:3], one)
