"""GPU: the token-split form of the attention kv launch (pcr_attn_params.kv_splits: partial per-cloud matrices + a fold
launch) gives the blocks' outputs of the single-launch form within summation order, for every split count and in both
arithmetic modes, and against the torch-eager oracle (reference models/pointnet2_utils.py:14-47,90-114; attention.py:192-219)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))

from pcr_amd import engine, testing as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,d,B,Lq,Sk", [("self", 32, 3, 300, 300), ("self", 64, 5, 512, 512), ("self", 128, 2, 256, 256),
                                            ("cross", 64, 4, 100, 1000), ("fp", 64, 3, 257, 131), ("cross", 32, 2, 64, 129)])
@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
def test_kv_token_split_equals_single_launch(kind, d, B, Lq, Sk, prec):
    import model_oracle as MO
    from mmdet3d.models.attention import corss_attention
    from mmdet3d.models.pointnet2_utils import FP_SA, Self_Attention
    g = torch.Generator().manual_seed(d + Lq)
    tt = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
    if kind == "self":
        m = Self_Attention(d, 2)
        args = (tt(B, d, Lq), tt(B, Lq, 3))
        oracle = MO.self_attention
    elif kind == "fp":
        m = FP_SA(0, 32, 128, d, 64, 2)
        args = (tt(B, 32, Lq), tt(B, Lq, 3), tt(B, 128, Sk), tt(B, Sk, 3))
        oracle = MO.fp_sa
    else:
        m = corss_attention(d, 2)
        args = (tt(B, d, Lq), tt(B, Lq, 3), tt(B, d, Sk), tt(B, Sk, 3))
        oracle = MO.cross_attention
    sd = T.seeded_state_dict(T.manifest_of(m), 3)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    with torch.no_grad():
        want = oracle(sd, *args)
    outs = {}
    prev = engine.KV_SPLITS
    try:
        with engine.precision(prec):
            for ns in (1, 2, 3, 4):
                engine.KV_SPLITS = ns
                with torch.no_grad():
                    outs[ns] = m(*[a.cuda() for a in args]).cpu()
                    again = m(*[a.cuda() for a in args]).cpu()
                assert torch.equal(outs[ns], again)                      # fixed summation order: run to run identical
    finally:
        engine.KV_SPLITS = prev
    tol = 1e-4
    for ns, o in outs.items():
        assert float((o - want).abs().max()) < tol, (ns, float((o - want).abs().max()))
        # f32: the forms differ by summation order only.  bf16x3 at d = c2 = 64 with whole 32-token blocks: the
        # single-launch form is the wave-autonomous kernel with a split-bf16 projection, the split form keeps the
        # tile kernel's f32 projection -- two arithmetics of the same product, both within the oracle tolerance
        same_arith = prec == "f32" or not (d in (32, 64) and args[-2].shape[1] == d and Sk % 32 == 0)
        lim = 2e-5 if same_arith else 1e-4
        assert float((o - outs[1]).abs().max()) < lim, (ns, float((o - outs[1]).abs().max()))


def test_split_suggestion_is_sane():
    from pcr_amd import _lib as L
    lib = L.load()
    assert lib.pcr_attn_kv_splits(1024, 1000, 64) == lib.pcr_attn_kv_splits(8, 1000, 64) == 4     # shape only, never B
    assert lib.pcr_attn_kv_splits(8, 1000, 32) == 4
    assert lib.pcr_attn_kv_splits(8, 1024, 32) == 1                 # whole blocks at d = 32: streaming too
    assert lib.pcr_attn_kv_splits(8, 500, 64) == 2
    assert lib.pcr_attn_kv_splits(8, 1024, 64) == 1                 # whole 32-token blocks at d = 64: the streaming kernel
    assert lib.pcr_attn_kv_splits(8, 64, 64) == 1                   # one tile per cloud: nothing to split
    assert lib.pcr_attn_kv_splits(8, 1024, 256) == 1                # the wide kernel splits by bands already
    assert lib.pcr_attn_kv_splits(4, 4096, 128) == 1
