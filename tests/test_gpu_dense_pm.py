"""GPU: the point-major-input dense launch on the bf16 matrix core (pcr_dense_xpm_prec_f32: SSG's Conv1d after the last
set-abstraction layer, models/pointnet2_ssg.py cov_final) against torch fp32 -- widths with a partly filled cout block,
token counts that leave a partial 32-token block, every activation, with and without scale / shift, both bf16 modes --
and against the f32 launch of the same call (pcr_dense_xpm_f32)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,cin,cout,L,act,affine", [(5, 256, 64, 128, 0, "shift"), (3, 64, 32, 50, 1, "both"),
                                                     (2, 128, 100, 33, 2, "both"), (1, 256, 128, 64, 0, "none"),
                                                     (300, 64, 7, 32, 1, "shift")])
def test_point_major_dense_on_the_bf16_core(B, cin, cout, L, act, affine):
    from pcr_amd import engine
    from pcr_amd import _lib as Lb
    g = torch.Generator().manual_seed(B + cin + cout + L)
    x_pm = torch.randn(B, L, cin, generator=g).cuda()          # point-major storage
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5)
    scale = (torch.rand(cout, generator=g) + 0.5).cuda() if affine == "both" else None
    shift = torch.randn(cout, generator=g).cuda() if affine in ("both", "shift") else None
    ref = torch.einsum("oc,blc->bol", w.double().cuda(), x_pm.double())
    if scale is not None:
        ref = ref * scale.double().view(1, -1, 1)
    if shift is not None:
        ref = ref + shift.double().view(1, -1, 1)
    ref = torch.relu(ref) if act == 1 else (torch.where(ref < 0, ref * 0.2, ref) if act == 2 else ref)
    wp = engine.pack_weight_dual(w, torch.device("cuda"))
    assert Lb.load().pcr_dense_xpm_prec_ok(cin, cout, L) == 1 and getattr(wp, "_pcr_bf", None) is not None
    view = x_pm.transpose(1, 2)                                # the (B,cin,L) view the model hands over
    den = float(ref.abs().max())
    out = {}
    for prec, tol in (("f32", 2e-6), ("bf16x3", 2e-5), ("bf16", 2e-2)):
        with engine.precision(prec):
            engine.PROFILE = []
            try:
                y = engine.dense(view, wp, cout, scale, shift, act)
                torch.cuda.synchronize()
                arith = engine.PROFILE[-1][-1]
            finally:
                engine.PROFILE = None
        assert y.shape == (B, cout, L) and y.is_contiguous()
        assert arith == prec, (arith, prec)                    # the launch that ran is the one asked for
        assert float((y.double() - ref).abs().max()) / den < tol, prec
        out[prec] = y
    assert float((out["bf16x3"] - out["f32"]).abs().max()) / den < 2e-5


def test_unsupported_shapes_keep_the_f32_launch():
    from pcr_amd import _lib as Lb
    lib = Lb.load()
    assert lib.pcr_dense_xpm_prec_ok(96, 64, 128) == 0          # cin not a multiple of 64
    assert lib.pcr_dense_xpm_prec_ok(256, 160, 128) == 0        # more than four cout blocks
    assert lib.pcr_dense_xpm_prec_ok(2048, 128, 128) == 0       # weight image beyond LDS
    assert lib.pcr_dense_xpm_prec_ok(512, 64, 1) == 1
