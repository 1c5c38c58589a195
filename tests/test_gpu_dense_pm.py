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


@pytest.mark.parametrize("B,cin,cout,L", [(6, 128, 128, 256), (4, 32, 128, 64), (3, 64, 64, 128), (2, 128, 64, 96), (1, 64, 128, 4096)])
@pytest.mark.parametrize("prec", ["bf16x3", "bf16"])
def test_layer1_tables_persistent_kernel_equals_the_tile_kernel_bit_for_bit(B, cin, cout, L, prec):
    """pcr_dense_pm_prec_f32 (the per-point tables of a set-abstraction layer's decomposed first conv) with point-major
    input: token counts B L that are whole 64-token tiles run dense_pm_res_kernel (round 5: persistent workgroups, weight
    rows resident in registers, the next tile requested behind the k-loop), any other count the one-shot tile kernel.
    A table row is a function of its own token and both kernels walk the 16-channel steps in the same order: the rows of
    a batch cut to a token count that is no multiple of 64 must carry the same bits.  And both stay on torch fp64."""
    from pcr_amd import engine as E, _lib as Lb
    g = torch.Generator().manual_seed(B + cin + cout + L)
    x = torch.randn(B, L, cin, generator=g).cuda()
    w = torch.randn(cout, cin, generator=g) / cin ** 0.5
    wp = E.pack_weight_bf(w, torch.device("cuda"))
    lib = Lb.load()

    def table(xs):
        b, l, _ = xs.shape
        y = torch.full((b, l, cout), float("nan"), device="cuda")
        Lb.check(lib.pcr_dense_pm_prec_f32(Lb.ptr(xs), Lb.ptr(wp), Lb.ptr(y), b, cin, cout, l, 1, E.PRECISIONS[prec],
                                           Lb.stream_ptr()), "pcr_dense_pm_prec_f32")
        return y

    full = table(x)
    assert (B * L) % 64 == 0 and (B * (L - 8)) % 64 != 0
    part = table(x[:, :L - 8].contiguous())
    assert torch.equal(full[:, :L - 8], part)
    ref = torch.einsum("oc,blc->blo", w.double().cuda(), x.double())
    tol = 2e-5 if prec == "bf16x3" else 2e-2
    assert float((full.double() - ref).abs().max()) / float(ref.abs().max()) < tol
