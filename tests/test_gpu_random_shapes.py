"""Randomly drawn shapes (hypothesis, derandomised: the same examples on every run) through the kernel
parity checks of tests/test_gpu_kernels.py -- ragged sizes, every padding / layout / fusion flag --
so that the hand-picked cases there are not the only ones that ever ran."""
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from tests import test_gpu_kernels as K

pytestmark = pytest.mark.gpu
ctx, O = K.ctx, K.O          # the module-scoped fixtures
CFG = dict(max_examples=30, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@settings(**CFG)
@given(batch=st.integers(1, 300), inf=st.integers(1, 260), outf=st.integers(1, 48), relu=st.integers(0, 1))
def test_linear_random_shapes(ctx, O, batch, inf, outf, relu):
    K.test_linear_fwd_bwd(ctx, O, batch, inf, outf, relu)


@settings(**CFG)
@given(ta=st.integers(0, 1), tb=st.integers(0, 1), m=st.integers(1, 200), n=st.integers(1, 200), k=st.integers(1, 700),
       ab=st.sampled_from([(1.0, 0.0), (1.0, 1.0), (0.5, -2.0)]))
def test_sgemm_random_shapes(ctx, O, ta, tb, m, n, k, ab):
    K.test_sgemm(ctx, O, (ta, tb, m, n, k), ab)


@settings(**CFG)
@given(n=st.integers(1, 6), c_in=st.integers(1, 40), h=st.integers(3, 24), w=st.integers(3, 30), c_out=st.integers(1, 70),
       pad=st.integers(0, 1), layout=st.integers(0, 1), relu=st.integers(0, 1))
def test_conv3x3_random_shapes(ctx, O, n, c_in, h, w, c_out, pad, layout, relu):
    K.test_conv3x3_fwd(ctx, O, n, c_in, h, w, c_out, pad, layout, relu)


@settings(**{**CFG, "max_examples": 15})
@given(n=st.integers(1, 40), c_in=st.integers(1, 24), hw=st.integers(3, 16), c_out=st.integers(1, 40), layout=st.integers(0, 1))
def test_conv3x3_backward_random_shapes(ctx, O, n, c_in, hw, c_out, layout):
    K.test_conv3x3_bwd_full_mode(ctx, O, n, c_in, hw, hw, c_out, layout)


@settings(**CFG)
@given(n=st.integers(1, 4), c=st.integers(1, 9), h=st.integers(3, 15), w=st.integers(3, 15), kh=st.integers(2, 3), kw=st.integers(2, 3),
       sh=st.integers(1, 3), sw=st.integers(1, 3), ph=st.integers(0, 1), pw=st.integers(0, 1))
def test_pools_random_geometry(ctx, O, n, c, h, w, kh, kw, sh, sw, ph, pw):
    if ph > kh // 2 or pw > kw // 2 or h + 2 * ph < kh or w + 2 * pw < kw:
        return                                   # not a valid pooling window (the reference asserts the same)
    # (windows of at least 2x2: the checked-in case plants a 2x2 NaN block, and a window that is ALL NaN without
    # containing element (0,0,0,0) is the one documented deviation of the gather-form backward)
    K.test_maxpool_bit_exact(ctx, O, n, c, h, w, (kh, kw), (sh, sw), (ph, pw))
    K.test_avgpool(ctx, O, n, c, h, w, (kh, kw), (sh, sw), (ph, pw))


@settings(**CFG)
@given(batch=st.integers(1, 400), classes=st.integers(1, 40))
def test_softmax_xent_random_shapes(ctx, O, batch, classes):
    K.test_softmax_xent(ctx, O, batch, classes)


@settings(**{**CFG, "max_examples": 40})
@given(n=st.integers(1, 40), c8=st.integers(1, 8), h=st.integers(1, 30), w=st.integers(1, 30), c_out=st.integers(1, 70), pad=st.integers(0, 1),
       relu=st.integers(0, 1))
def test_image_resident_conv_random_shapes(ctx, O, n, c8, h, w, c_out, pad, relu):
    """the image-resident matrix-core kernel FORCED (it normally takes only chip-filling launches) onto random geometries: ragged image
    groups, partial pixel tiles, channel counts that are no multiple of 16, pad 0 -- against the oracle; and, where the shape allows, the
    fused 2x2 max-pool and global-average-pool epilogues against the unfused results"""
    from tests import test_gpu_full_size as F
    if h + 2 * pad < 3 or w + 2 * pad < 3:
        return
    c_in = 8 * c8
    ctx.call("th_debug_set_conv_img", 1)
    try:
        K.test_conv3x3_fwd(ctx, O, n, c_in, h, w, c_out, pad, 0, relu)
        cfg = F.last_conv_config(ctx)
    finally:
        ctx.call("th_debug_set_conv_img", -1)
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    if 8 * (ho + 2) * (wo + 2) <= 8192:                       # one image's patch fits the staging plan: the image kernel must have run
        assert cfg["dma"] in (2, 3, 4, 5), cfg
    if c_out % 4 == 0 and F.ctx_supported_gap(n, c_in, h, w, c_out, pad):
        F.gap_case(ctx, O, n, c_in, h, w, c_out, pad)


@settings(**{**CFG, "max_examples": 20})
@given(b16=st.integers(1, 12), in16=st.integers(1, 20), h1_16=st.integers(1, 10), h2_16=st.integers(1, 10), c=st.integers(1, 16), need_dx=st.booleans())
def test_mlp3_random_shapes(ctx, O, b16, in16, h1_16, h2_16, c, need_dx):
    """th_mlp3_xent on random multiples of 16 (the runtime-size instance, with and without the first layer as its own launch, with and
    without dX) against the oracle's tape"""
    from tests import test_gpu_mlp3 as M
    M.test_mlp3_matches_the_oracle_tape(ctx, O, 16 * b16, 16 * in16, 16 * h1_16, 16 * h2_16, c, need_dx)


@settings(**{**CFG, "max_examples": 6})
@given(n=st.integers(1, 70), name=st.sampled_from(["reference", "simple"]))
def test_conv_chain_random_batches(ctx, O, n, name):
    """th_conv_chain_fwd at random batch sizes (the grid is one workgroup per image: any n) against the oracle's layer-by-layer ops"""
    from tests import test_gpu_chain as CH
    CH.test_chain_matches_the_oracle(ctx, O, name, n)


# ---- the large-batch MLP step on drawn shapes: ragged hidden widths (any multiple of 4), ragged batches, few / many classes -------------------
from tests import test_gpu_mlp2 as M2   # noqa: E402

m2_ctx, m2_O = M2.ctx, M2.O


@settings(**{**CFG, "max_examples": 24})
@given(batch=st.integers(32, 900), inf4=st.integers(8, 60), hid4=st.integers(1, 32), c=st.integers(1, 16))
def test_mlp2_random_shapes(m2_ctx, m2_O, batch, inf4, hid4, c):
    M2.test_mlp2_dense_rows(m2_ctx, m2_O, batch, 4 * inf4, 4 * hid4, c)


@settings(**{**CFG, "max_examples": 16})
@given(batch=st.integers(32, 700), inf4=st.integers(8, 50), h1=st.integers(1, 32), h2=st.integers(1, 32), c=st.integers(1, 16))
def test_mlp2_deep_random_shapes(m2_ctx, m2_O, batch, inf4, h1, h2, c):
    M2.test_mlp2_deep_dense_rows(m2_ctx, m2_O, batch, 4 * inf4, 4 * h1, 4 * h2, c)
