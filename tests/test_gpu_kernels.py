"""Per-kernel parity: every entry point of include/taper_hip.h on the hot path
is called through the C ABI on the GPU and compared with the CPU oracle on the
same seeded inputs.  Bar: bit-exact for index / argmax / pool-index work,
fp32 within 1e-4 relative (north_star) for everything else."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(scope="module")
def ctx():
    from taper_amd import hip
    c = hip.Ctx(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def close(a, b, rtol=RTOL, atol=1e-6):
    """relative + absolute criterion; the scale of the reference sets the floor near zero"""
    b = np.asarray(b)
    scale = float(np.abs(b).max()) if b.size else 0.0
    np.testing.assert_allclose(np.asarray(a).reshape(b.shape), b, rtol=rtol, atol=atol + rtol * 1e-2 * scale)


# ------------------------------------------------------------------ sgemm
MLP_SHAPES = [  # (ta, tb, m, n, k): the hot-path GEMMs of BASELINE configs
    (0, 0, 64, 128, 784), (0, 0, 64, 10, 128), (1, 0, 784, 128, 64), (1, 0, 128, 10, 64), (0, 1, 64, 128, 10),
    (0, 1, 128, 784, 10), (0, 0, 256, 128, 128), (0, 0, 1024, 10, 64),
]
ODD_SHAPES = [(0, 0, 1, 1, 1), (0, 0, 2, 3, 2), (1, 1, 17, 5, 33), (0, 1, 33, 65, 129), (1, 0, 130, 70, 9),
              (0, 0, 16, 16, 2000), (1, 0, 40, 24, 4100), (0, 0, 3, 200, 7)]
BIG_SHAPES = [(0, 0, 1024, 1024, 256), (0, 1, 1024, 1152, 96), (1, 0, 1280, 1024, 64), (1, 1, 1024, 1024, 40),
              (0, 0, 1100, 1030, 70),
              # deep K on a narrow output: split-K slices of the 128x128 kernel + the fixed-order reduce
              (1, 0, 128, 784, 4096), (1, 0, 128, 784, 5000), (0, 1, 4096, 128, 784), (0, 0, 256, 256, 3000), (1, 1, 130, 200, 2050),
              # ragged weight gradients (dW = dZ^T X with in = 784, out = 256 / 200 / 132): LDS-DMA with the edge quads zeroed by the descriptor's
              # range and guarded stores (sgemm_tile<.., RAG>, r06); 1300: m % 4 != 0 keeps the clamped register loads
              (1, 0, 784, 256, 4096), (1, 0, 256, 784, 2048), (1, 0, 784, 200, 1024), (1, 0, 132, 784, 8192), (1, 0, 1300, 1026, 512),
              # ... and k-contiguous operands: rows past m / n, and in the last k chunk the quads past k (forward X W^T and dX = dZ W at in = 784)
              (0, 1, 2048, 256, 784), (0, 1, 1000, 260, 784), (0, 0, 2048, 784, 256), (0, 0, 1100, 784, 260), (1, 1, 1000, 1028, 200),
              (0, 1, 16384, 256, 784)]


# (alpha, beta) = (0.5, -2.0) on every shape but the big ones (> 5e7 multiply-adds): two variants of those are enough
SGEMM_CASES = [(s, ab) for ab in [(1.0, 0.0), (1.0, 1.0), (0.5, -2.0)] for s in MLP_SHAPES + ODD_SHAPES + BIG_SHAPES
               if not (s[2] * s[3] * s[4] > 5e7 and ab == (0.5, -2.0))]


@pytest.mark.parametrize("shape,ab", SGEMM_CASES, ids=[f"{ab[0]}-{ab[1]}-" + "-".join(map(str, s)) for s, ab in SGEMM_CASES])
def test_sgemm(ctx, O, shape, ab):
    (ta, tb, m, n, k), (alpha, beta) = shape, ab
    rng = np.random.default_rng(m * 31 + n * 7 + k)
    a = rng.uniform(-1, 1, (k, m) if ta else (m, k)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, k) if tb else (k, n)).astype(np.float32)
    c0 = rng.uniform(-1, 1, (m, n)).astype(np.float32)
    ref = c0.copy()
    O.sgemm_rowmajor(ta, tb, m, n, k, alpha, a, b, beta, ref)
    da, db, dc = ctx.upload(a), ctx.upload(b), ctx.upload(c0)
    ctx.call("th_sgemm", ta, tb, m, n, k, alpha, da, db, beta, dc)
    got = ctx.download(dc, (m, n))
    # fp32 accumulation-order tolerance scales with sqrt(k)
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=2e-6 * np.sqrt(k) * max(1.0, abs(alpha)) + 1e-6)


def test_sgemm_beta0_ignores_nan_c(ctx):
    """beta == 0 overwrites C without reading it (matrixmultiply / gemm.rs semantics)"""
    a = np.ones((32, 8), np.float32)
    b = np.ones((8, 32), np.float32)
    c = np.full((32, 32), np.nan, np.float32)
    dc = ctx.upload(c)
    ctx.call("th_sgemm", 0, 0, 32, 32, 8, 1.0, ctx.upload(a), ctx.upload(b), 0.0, dc)
    np.testing.assert_array_equal(ctx.download(dc, (32, 32)), np.full((32, 32), 8.0, np.float32))


def test_sgemm_transpose_detecting(ctx):
    """A = I with an ASYMMETRIC B catches a swapped C write (guide: always A=I-check)"""
    n = 256
    a = np.eye(n, dtype=np.float32)
    b = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 97) + np.arange(n, dtype=np.float32)[:, None] * 0.5
    dc = ctx.empty(n * n)
    ctx.call("th_sgemm", 0, 0, n, n, n, 1.0, ctx.upload(a), ctx.upload(b), 0.0, dc)
    np.testing.assert_array_equal(ctx.download(dc, (n, n)), b)
    big = 1024  # through the 128x128 MFMA kernel
    a = np.eye(big, dtype=np.float32)
    b = ((np.arange(big * big, dtype=np.int64).reshape(big, big) * 7919) % 1013).astype(np.float32)
    dc = ctx.empty(big * big)
    ctx.call("th_sgemm", 0, 0, big, big, big, 1.0, ctx.upload(a), ctx.upload(b), 0.0, dc)
    np.testing.assert_array_equal(ctx.download(dc, (big, big)), b)


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0)])
def test_sgemm_4096_sampled_rows(ctx, ta, tb):
    """BASELINE configs[4] size; the oracle is too slow at 137 GFLOP, so 48 random
    rows are checked against float64 numpy and the rest through linearity."""
    n = 4096
    rng = np.random.default_rng(4096 + ta * 2 + tb)
    a = rng.uniform(-1, 1, (n, n)).astype(np.float32)
    b = rng.uniform(-1, 1, (n, n)).astype(np.float32)
    da, db, dc = ctx.upload(a), ctx.upload(b), ctx.empty(n * n)
    ctx.call("th_sgemm", ta, tb, n, n, n, 1.0, da, db, 0.0, dc)
    got = ctx.download(dc, (n, n))
    rows = rng.choice(n, 48, replace=False)
    opa = (a.T if ta else a)[rows].astype(np.float64)
    opb = (b.T if tb else b).astype(np.float64)
    np.testing.assert_allclose(got[rows], opa @ opb, rtol=RTOL, atol=2e-4)
    # linearity: C(2A, B) with beta=-1 on top of the first result gives C again
    da2 = ctx.upload(2 * a)
    ctx.call("th_sgemm", ta, tb, n, n, n, 1.0, da2, db, -1.0, dc)
    np.testing.assert_allclose(ctx.download(dc, (n, n)), got, rtol=RTOL, atol=2e-4)


# ------------------------------------------------------------------ linear
@pytest.mark.parametrize("batch,inf,outf", [(64, 784, 128), (64, 128, 10), (32, 784, 128), (1, 5, 3), (128, 128, 64), (256, 3136, 10),
                                            (4096, 784, 128), (5000, 200, 130),   # large batch: tile kernels, split-K dW
                                            (1024, 784, 128), (700, 100, 30)])     # mid batch: K slices inside the one-launch backward
@pytest.mark.parametrize("relu", [0, 1])
def test_linear_fwd_bwd(ctx, O, batch, inf, outf, relu):
    rng = np.random.default_rng(batch + inf + outf)
    x = rng.uniform(0, 1, (batch, inf)).astype(np.float32)
    w = rng.uniform(-0.1, 0.1, (outf, inf)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, outf).astype(np.float32)
    dy = rng.uniform(-1, 1, (batch, outf)).astype(np.float32)
    # oracle: the reference's three-node chain (nn.rs:54-60)
    O.Tape.reset()
    O.Tape.set_zero_sentinel(False)
    xt, wt, bt = O.Tensor(x).requires_grad(), O.Tensor(w).requires_grad(), O.Tensor(b).requires_grad()
    pre = xt.matmul(wt.transpose()).add_broadcast(bt)
    out = pre.relu() if relu else pre
    (out * O.Tensor(dy)).sum(None, False).backward()
    O.Tape.set_zero_sentinel(True)
    dx_, dw_, db_, dyd = ctx.upload(x), ctx.upload(w), ctx.upload(b), ctx.upload(dy)
    y = ctx.empty(batch * outf)
    ctx.call("th_linear_fwd", dx_, dw_, db_, y, batch, inf, outf, relu)
    yv = ctx.download(y, (batch, outf))
    close(yv, out.data())
    gx, gw, gb = ctx.empty(batch * inf), ctx.empty(outf * inf), ctx.empty(outf)
    # the ReLU backward is folded into the kernel through the post-activation output (Q15)
    ctx.call("th_linear_bwd", dx_, dw_, dyd, y if relu else None, gx, gw, gb, batch, inf, outf, 0)   # grads were None: overwrite
    deep = dict(atol=3e-8 * batch)   # dW / db add `batch` terms of O(1): absolute error of a reordered fp32 sum
    close(ctx.download(gx, (batch, inf)), xt.grad())
    close(ctx.download(gw, (outf, inf)), wt.grad(), **deep)
    close(ctx.download(gb, (outf,)), bt.grad(), **deep)
    ctx.call("th_linear_bwd", dx_, dw_, dyd, y if relu else None, gx, gw, gb, batch, inf, outf, 7)   # accumulate on top
    close(ctx.download(gw, (outf, inf)), 2 * wt.grad(), **deep)
    close(ctx.download(gx, (batch, inf)), 2 * xt.grad())
    close(ctx.download(gb, (outf,)), 2 * bt.grad(), **deep)
    # partial outputs: dW only (layer 1 of the MLP: the input has no grad, ops.rs:243)
    gw2 = ctx.empty(outf * inf)
    ctx.call("th_linear_bwd", dx_, None, dyd, y if relu else None, None, gw2, None, batch, inf, outf, 0)
    close(ctx.download(gw2, (outf, inf)), wt.grad(), **deep)
    O.Tape.reset()


@pytest.mark.parametrize("batch,inf,outf", [(2816, 2816, 256), (2800, 2900, 300), (512, 2048, 10), (700, 2048, 16), (4096, 4096, 10)])
def test_linear_bwd_dx_epilogue(ctx, batch, inf, outf):
    """th_linear_bwd_adam_ex2: the backward of the fused Linear + ReLU in front (ops.rs:358-369; tensor.rs:686-691) folded into this layer's
    dX product -- d_dx * [d_x > 0] bit-identical to the plain call's d_dx masked afterwards, the row blocks' column sums of it adding up to
    the column sums of that, dW / db untouched -- on the unsplit 128-tiles (whole and ragged) and on the thin layer's streaming kernel"""
    from taper_amd._lib import hip as lib
    rows = lib.th_linear_bwd_dx_epilogue_rows(batch, inf, outf)
    assert rows > 0
    assert lib.th_linear_bwd_dx_epilogue_rows(64, 784, 128) == 0            # the latency-bound shapes keep the one-launch backward
    rng = np.random.default_rng(batch + inf + outf)
    x = np.maximum(rng.standard_normal((batch, inf)), 0).astype(np.float32)  # a post-ReLU input: about half of it zeros
    w = rng.uniform(-0.1, 0.1, (outf, inf)).astype(np.float32)
    dy = rng.uniform(-1, 1, (batch, outf)).astype(np.float32)
    dx_, dw_, dyd = ctx.upload(x), ctx.upload(w), ctx.upload(dy)
    gx0, gw0, gb0 = ctx.empty(batch * inf), ctx.empty(outf * inf), ctx.empty(outf)
    ctx.call("th_linear_bwd_adam_ex", dx_, dw_, dyd, None, gx0, gw0, gb0, batch, inf, outf, 0, None, None, None, 0)
    gx1, gw1, gb1, part = ctx.empty(batch * inf), ctx.empty(outf * inf), ctx.empty(outf), ctx.empty(rows * inf)
    ctx.call("th_linear_bwd_adam_ex2", dx_, dw_, dyd, None, gx1, gw1, gb1, batch, inf, outf, 0, None, None, None, 0, 1, part)
    plain = ctx.download(gx0, (batch, inf))
    want = np.where(x > 0, plain, np.float32(0))
    if outf > 16:
        np.testing.assert_array_equal(ctx.download(gx1, (batch, inf)), want)    # the same product, masked in its epilogue
    else:                                                                       # (the thin kernel adds k in order with fmaf: another rounding than the tiles')
        ref = np.where(x > 0, dy.astype(np.float64) @ w.astype(np.float64), 0.0)
        np.testing.assert_allclose(ctx.download(gx1, (batch, inf)), ref, rtol=RTOL, atol=1e-6)
        np.testing.assert_allclose(plain, dy.astype(np.float64) @ w.astype(np.float64), rtol=RTOL, atol=1e-6)
    np.testing.assert_array_equal(ctx.download(gw1, (outf, inf)), ctx.download(gw0, (outf, inf)))
    np.testing.assert_array_equal(ctx.download(gb1, outf), ctx.download(gb0, outf))
    cs = ctx.empty(inf)
    ctx.call("th_colsum", part, cs, rows, inf)
    got_masked = ctx.download(gx1, (batch, inf)).astype(np.float64)
    np.testing.assert_allclose(ctx.download(cs, inf), got_masked.sum(axis=0), rtol=RTOL, atol=1e-5 * float(np.abs(got_masked.sum(axis=0)).max()))
    # without the mask: column partials of the plain product
    ctx.call("th_linear_bwd_adam_ex2", dx_, dw_, dyd, None, gx1, None, None, batch, inf, outf, 0, None, None, None, 0, 0, part)
    ctx.call("th_colsum", part, cs, rows, inf)
    p64 = ctx.download(gx1, (batch, inf)).astype(np.float64)
    np.testing.assert_allclose(ctx.download(cs, inf), p64.sum(axis=0), rtol=RTOL, atol=1e-5 * float(np.abs(p64.sum(axis=0)).max()))


# ------------------------------------------------------------------ element-wise
@pytest.mark.parametrize("n", [1, 3, 4, 1000, 100_003, 1 << 20])
def test_elementwise(ctx, n):
    rng = np.random.default_rng(n)
    a = rng.uniform(-2, 2, n).astype(np.float32)
    b = rng.uniform(0.5, 2, n).astype(np.float32)
    g = rng.uniform(-1, 1, n).astype(np.float32)
    da, db, dg, out = ctx.upload(a), ctx.upload(b), ctx.upload(g), ctx.empty(n)
    for name, ref in [("th_add", a + b), ("th_sub", a - b), ("th_mul", a * b), ("th_div", a / b)]:
        ctx.call(name, da, db, out, n)
        np.testing.assert_array_equal(ctx.download(out, n), ref)  # IEEE basic ops: bit-exact
    ctx.call("th_relu_fwd", da, out, n)
    np.testing.assert_array_equal(ctx.download(out, n), np.maximum(a, 0))
    acc0 = rng.uniform(-1, 1, n).astype(np.float32)
    acc = ctx.upload(acc0)
    ctx.call("th_relu_bwd", da, dg, acc, n, 1)
    np.testing.assert_array_equal(ctx.download(acc, n), acc0 + np.where(a > 0, g, 0).astype(np.float32))
    ctx.call("th_relu_bwd", da, dg, acc, n, 0)
    np.testing.assert_array_equal(ctx.download(acc, n), np.where(a > 0, g, 0).astype(np.float32))
    acc = ctx.upload(acc0)
    ctx.call("th_axpy", 1.0, dg, acc, n)
    np.testing.assert_array_equal(ctx.download(acc, n), acc0 + g)
    ctx.call("th_axpy", -1.0, dg, acc, n)
    np.testing.assert_array_equal(ctx.download(acc, n), (acc0 + g) + np.float32(-1.0) * g)
    acc = ctx.upload(acc0)
    ctx.call("th_mul_bwd", dg, db, acc, n)
    np.testing.assert_array_equal(ctx.download(acc, n), acc0 + g * b)
    # transcendental ops: device libm vs host libm within a few ulp
    ctx.call("th_exp_fwd", da, out, n)
    np.testing.assert_allclose(ctx.download(out, n), np.exp(a), rtol=1e-6)
    ctx.call("th_log_fwd", db, out, n)
    np.testing.assert_allclose(ctx.download(out, n), np.log(b), rtol=1e-6, atol=1e-7)
    ctx.call("th_sigmoid_fwd", da, out, n)
    sig = np.where(a > 0, 1 / (1 + np.exp(-a.astype(np.float64))), np.exp(a.astype(np.float64)) / (1 + np.exp(a.astype(np.float64))))
    np.testing.assert_allclose(ctx.download(out, n), sig, rtol=1e-6)
    s = ctx.download(out, n)
    acc = ctx.upload(acc0)
    ctx.call("th_sigmoid_bwd", out, dg, acc, n)
    np.testing.assert_allclose(ctx.download(acc, n), acc0 + g * s * (1 - s), rtol=1e-6, atol=1e-7)
    ctx.call("th_pow_fwd", db, 0.5, out, n)
    np.testing.assert_allclose(ctx.download(out, n), np.sqrt(b), rtol=1e-6)
    acc = ctx.upload(acc0)
    ga, gb = ctx.upload(acc0), ctx.upload(acc0)
    ctx.call("th_div_bwd", dg, da, db, ga, gb, n)
    np.testing.assert_allclose(ctx.download(ga, n), acc0 + g / b, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ctx.download(gb, n), acc0 - g * a / (b * b), rtol=1e-5, atol=1e-6)


def test_relu_nan_matches_maxps(ctx):
    """_mm_max_ps(v, 0) returns 0 for NaN (ops.rs:328); -0.0 -> +0.0"""
    a = np.array([np.nan, -0.0, 0.0, -1.0, 2.0, np.inf, -np.inf, 1e-45], np.float32)
    out = ctx.empty(a.size)
    ctx.call("th_relu_fwd", ctx.upload(a), out, a.size)
    got = ctx.download(out, a.size)
    np.testing.assert_array_equal(got, np.array([0, 0, 0, 0, 2, np.inf, 0, 1e-45], np.float32))
    assert not np.signbit(got[1])


# ------------------------------------------------------------------ layout / broadcast / reduce
@pytest.mark.parametrize("rows,cols", [(128, 784), (10, 128), (1, 7), (65, 63), (4096, 512)])
def test_transpose(ctx, rows, cols):
    rng = np.random.default_rng(rows * cols)
    x = rng.uniform(-1, 1, (rows, cols)).astype(np.float32)
    out = ctx.empty(rows * cols)
    ctx.call("th_transpose2d", ctx.upload(x), out, rows, cols)
    np.testing.assert_array_equal(ctx.download(out, (cols, rows)), x.T)
    gout = rng.uniform(-1, 1, (cols, rows)).astype(np.float32)
    gin0 = rng.uniform(-1, 1, (rows, cols)).astype(np.float32)
    gin = ctx.upload(gin0)
    ctx.call("th_transpose2d_bwd", ctx.upload(gout), gin, rows, cols)
    np.testing.assert_array_equal(ctx.download(gin, (rows, cols)), gin0 + gout.T)


@pytest.mark.parametrize("rows,cols", [(64, 128), (64, 10), (1, 1), (1000, 10), (7, 300), (4096, 4096), (16384, 128), (20001, 10),
                                       (1024, 70)])   # tall: row slabs + a second pass
def test_bias_colsum_rowops(ctx, O, rows, cols):
    rng = np.random.default_rng(rows + cols)
    x = rng.uniform(-1, 1, (rows, cols)).astype(np.float32)
    bias = rng.uniform(-1, 1, cols).astype(np.float32)
    r = rng.uniform(-1, 1, rows).astype(np.float32)
    dx, dbias, dr, out = ctx.upload(x), ctx.upload(bias), ctx.upload(r), ctx.empty(rows * cols)
    ctx.call("th_bias_add_rows", dx, dbias, out, rows, cols, 0)
    np.testing.assert_array_equal(ctx.download(out, (rows, cols)), x + bias)
    ctx.call("th_bias_add_rows", dx, dbias, out, rows, cols, 1)
    np.testing.assert_array_equal(ctx.download(out, (rows, cols)), np.maximum(x + bias, 0))
    ctx.call("th_sub_rows", dx, dr, out, rows, cols)
    np.testing.assert_array_equal(ctx.download(out, (rows, cols)), x - r[:, None])
    # sums: order differs from the reference's sequential loop -> tolerance, float64 truth
    x64 = x.astype(np.float64)
    tol = dict(rtol=RTOL, atol=1e-6 * max(rows, cols))
    gb0 = rng.uniform(-1, 1, cols).astype(np.float32)
    gb = ctx.upload(gb0)
    ctx.call("th_colsum_accum", dx, gb, rows, cols)
    np.testing.assert_allclose(ctx.download(gb, cols), gb0 + x64.sum(0), **tol)
    ctx.call("th_colsum", dx, gb, rows, cols)
    np.testing.assert_allclose(ctx.download(gb, cols), x64.sum(0), **tol)
    rs = ctx.empty(rows)
    ctx.call("th_rowsum", dx, rs, rows, cols)
    np.testing.assert_allclose(ctx.download(rs, rows), x64.sum(1), **tol)
    gr = ctx.upload(r)
    ctx.call("th_rowsum_neg_accum", dx, gr, rows, cols)
    np.testing.assert_allclose(ctx.download(gr, rows), r - x64.sum(1), **tol)
    s = ctx.empty(1)
    ctx.call("th_sum_all", dx, s, rows * cols, 1.0)
    np.testing.assert_allclose(ctx.download(s, 1)[0], x64.sum(), rtol=RTOL, atol=1e-6 * rows * cols ** 0.5)
    ctx.call("th_sum_all", dx, s, rows * cols, float(rows * cols))
    np.testing.assert_allclose(ctx.download(s, 1)[0], x64.mean(), rtol=RTOL, atol=1e-6)
    g0 = rng.uniform(-1, 1, (rows, cols)).astype(np.float32)
    gin = ctx.upload(g0)
    ctx.call("th_rowsum_bwd", dr, gin, rows, cols)
    np.testing.assert_array_equal(ctx.download(gin, (rows, cols)), g0 + r[:, None])
    gin = ctx.upload(g0)
    ctx.call("th_colsum_bwd", dbias, gin, rows, cols)
    np.testing.assert_array_equal(ctx.download(gin, (rows, cols)), g0 + bias)


@pytest.mark.parametrize("rows,cols", [(64, 10), (3, 4), (1, 1), (1024, 10), (5, 1000), (2, 64), (2, 65)])
def test_rowmax_colmax_bit_exact(ctx, O, rows, cols):
    rng = np.random.default_rng(rows * 1000 + cols)
    # few distinct values -> many ties: the first maximum must win (tensor.rs:1062)
    x = rng.integers(0, 4, (rows, cols)).astype(np.float32)
    if rows > 2 and cols > 2:
        x[1, :] = np.nan      # all-NaN row -> value -inf, index 0
        x[2, 0] = np.nan      # NaN never wins
    xt = O.Tensor(x)
    v, i = xt.max(1)
    dv, di = ctx.empty(rows), ctx.empty(rows)
    ctx.call("th_rowmax", ctx.upload(x), dv, di, rows, cols)
    np.testing.assert_array_equal(ctx.download(di, rows), i.data().reshape(-1))
    np.testing.assert_array_equal(ctx.download(dv, rows), v.data().reshape(-1))
    v, i = xt.max(0)
    dv, di = ctx.empty(cols), ctx.empty(cols)
    ctx.call("th_colmax", ctx.upload(x), dv, di, rows, cols)
    np.testing.assert_array_equal(ctx.download(di, cols), i.data().reshape(-1))
    np.testing.assert_array_equal(ctx.download(dv, cols), v.data().reshape(-1))


@pytest.mark.parametrize("n", [1, 2, 7, 64, 65, 4096, 4097, 300_000, 5_000_000])
def test_global_max_last_of_equal_maxima(ctx, O, n):
    """tensor.rs:1072-1083: `max_by(partial_cmp)` keeps the LAST of equal maxima; index as f32; bit-exact vs the oracle"""
    rng = np.random.default_rng(n)
    x = rng.integers(0, 3, n).astype(np.float32)       # three distinct values -> the maximum repeats ~n/3 times
    v, i = O.Tensor(x).max(None)
    dv, di, df = ctx.empty(1), ctx.empty(1), ctx.empty(1)
    ctx.call("th_global_max", ctx.upload(x), n, dv, di, df)
    assert ctx.download(dv, 1)[0] == v.data()[0]
    assert ctx.download(di, 1)[0] == i.data()[0]
    last = np.float32(np.flatnonzero(x == x.max())[-1])
    assert ctx.download(di, 1)[0] == last
    assert ctx.download(df, 1).view(np.int32)[0] == 0
    # the host mirror (Tensor::max(None) / argmax(None)) takes the same kernel
    import taper_amd as T
    hv, hi = T.Tensor(x).max(None)
    assert hv.data()[0] == v.data()[0] and hi.data()[0] == i.data()[0]
    assert T.Tensor(x).argmax(None).data()[0] == i.data()[0]


def test_global_max_nan_panics_like_partial_cmp_unwrap(ctx):
    import taper_amd as T
    x = np.array([1.0, np.nan, 3.0], np.float32)
    with pytest.raises(Exception, match="unwrap"):
        T.Tensor(x).max(None)
    v, i = T.Tensor(np.array([np.nan], np.float32)).max(None)   # one element: no comparison, no panic
    assert np.isnan(v.data()[0]) and i.data()[0] == 0.0


# ------------------------------------------------------------------ softmax cross-entropy
@pytest.mark.parametrize("batch,classes", [(64, 10), (128, 10), (256, 10), (1024, 10), (1, 2), (7, 3), (32, 100), (5, 1),
                                           (2048, 10), (1500, 37), (63, 16), (65, 17)])
def test_softmax_xent(ctx, O, batch, classes):
    rng = np.random.default_rng(batch * 17 + classes)
    logits = (rng.standard_normal((batch, classes)) * 3).astype(np.float32)
    if batch > 4:
        logits[3] = 1000.0 + np.arange(classes)        # tests/smoke.rs:504-523 stability case
        logits[4] = np.round(logits[4])                # ties for the argmax
    targets = rng.integers(0, classes, batch).astype(np.float32)
    O.Tape.reset()
    lt = O.Tensor(logits).requires_grad()
    tt = O.Tensor(targets)
    loss = O.cross_entropy_loss(lt, tt)
    O.Tape.set_zero_sentinel(False)
    loss.backward()
    O.Tape.set_zero_sentinel(True)
    ref_logp = O.log_softmax(O.Tensor(logits)).data()
    ref_am = O.Tensor(logits).argmax(1).data().reshape(-1)
    ref_acc = O.accuracy(O.Tensor(logits), tt)
    dl, dt = ctx.upload(logits), ctx.upload(targets)
    logp, dloss, am, nc = ctx.empty(batch * classes), ctx.empty(1), ctx.empty(batch), ctx.empty(1)
    dunit = ctx.empty(batch * classes)
    state, metrics = ctx.upload(np.array([2, 100], np.int64)), ctx.zeros(2 * 8)
    ctx.call("th_softmax_xent_fwd", dl, dt, batch, classes, logp, dloss, am, nc, dunit, metrics, 8, state, batch, None)
    close(ctx.download(dunit, (batch, classes)), lt.grad(), atol=1e-7)      # unit-upstream gradient from the forward kernel
    np.testing.assert_array_equal(ctx.download(state, 2, np.int64), [3, 100 + batch])   # fused th_log_step
    mrow = ctx.download(metrics, (8, 2))[2]
    assert mrow[0] == ctx.download(dloss, 1)[0] and mrow[1] == ctx.download(nc, 1)[0]
    ctx.call("th_softmax_xent_fwd", dl, dt, batch, classes, None, dloss, None, None, None, None, 0, None, 0, None)   # all optionals off
    np.testing.assert_allclose(ctx.download(dloss, 1)[0], loss.data()[0], rtol=RTOL, atol=1e-6)
    close(ctx.download(logp, (batch, classes)), ref_logp, atol=1e-5)
    np.testing.assert_allclose(ctx.download(dloss, 1)[0], loss.data()[0], rtol=RTOL, atol=1e-6)
    np.testing.assert_array_equal(ctx.download(am, batch), ref_am)                 # index work: bit-exact
    assert ctx.download(nc, 1)[0] == round(ref_acc * batch)
    g0 = ctx.upload(np.ones(1, np.float32))
    dlog = ctx.empty(batch * classes)
    ctx.call("th_softmax_xent_bwd", logp, dt, g0, batch, classes, dlog, 0)
    close(ctx.download(dlog, (batch, classes)), lt.grad(), atol=1e-7)
    ctx.call("th_softmax_xent_bwd", logp, dt, g0, batch, classes, dlog, 1)
    close(ctx.download(dlog, (batch, classes)), 2 * lt.grad(), atol=1e-7)
    lp2 = ctx.empty(batch * classes)
    ctx.call("th_log_softmax_fwd", dl, lp2, batch, classes)
    close(ctx.download(lp2, (batch, classes)), ref_logp, atol=1e-5)
    cnt = ctx.empty(1)
    ctx.call("th_accuracy_count", am, dt, batch, cnt)
    assert ctx.download(cnt, 1)[0] == round(ref_acc * batch)
    O.Tape.reset()


# ------------------------------------------------------------------ conv / pool
CONV_CASES = [  # n, c_in, h, w, c_out, pad  (reference CNN layers at a small batch + edge cases)
    (4, 1, 28, 28, 32, 1), (3, 32, 28, 28, 32, 1), (3, 32, 14, 14, 64, 1), (2, 64, 14, 14, 64, 1), (6, 64, 7, 7, 128, 1),
    (1, 3, 5, 5, 2, 1), (2, 5, 9, 6, 7, 1), (2, 4, 8, 8, 20, 0), (1, 1, 3, 3, 1, 0), (7, 9, 7, 7, 17, 1),
    # matrix-core path (C_in >= 8): ragged channel counts, pad 0, wide / tall / tiny planes, > 128 output channels
    (5, 8, 10, 12, 16, 1), (3, 13, 9, 11, 33, 0), (2, 16, 3, 40, 8, 1), (9, 24, 5, 5, 150, 1), (2, 40, 30, 30, 48, 0),
    (3, 8, 3, 3, 4, 0), (1, 64, 1, 1, 10, 1), (1, 8, 20, 120, 16, 1), (1, 8, 6, 130, 8, 1),   # widest matrix-core plane / VALU fallback
]


@pytest.mark.parametrize("n,c_in,h,w,c_out,pad", CONV_CASES)
@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("relu", [0, 1])
def test_conv3x3_fwd(ctx, O, n, c_in, h, w, c_out, pad, layout, relu):
    rng = np.random.default_rng(n + c_in * 3 + h * 5 + c_out * 7 + pad)
    x = rng.uniform(-1, 1, (n, c_in, h, w)).astype(np.float32)
    wt = rng.uniform(-0.5, 0.5, (c_out, c_in, 3, 3)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, c_out).astype(np.float32)
    xt, wtt, bt = O.Tensor(x), O.Tensor(wt), O.Tensor(b)
    if layout == 0:   # the live reference path: im2col + reinterpreted weight (tensor.rs:1221-1285, Q3)
        ref = (xt.conv2d_relu if relu else xt.conv2d)(wtt, bt, (1, 1), (pad, pad), (1, 1))
    else:             # the reference's direct kernel with standard weights (tensor.rs:1287-1376)
        ref = xt.conv2d_direct_3x3(wtt, bt, (1, 1), (pad, pad))
        ref = ref.relu() if relu else ref
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    # The dead-code direct kernel sizes its output (h + 2p - 2)/s + 1 (tensor.rs:1307-1308), one
    # row/column larger than a 3x3 convolution produces; the extra border holds partial sums.
    # The live path and th_conv3x3_fwd use the conv2d geometry (tensor.rs:1254-1255): compare there.
    assert ref.shape() == ((n, c_out, ho, wo) if layout == 0 else (n, c_out, ho + 1, wo + 1))
    ref_d = ref.data()[:, :, :ho, :wo]
    y = ctx.empty(n * c_out * ho * wo)
    ctx.call("th_conv3x3_fwd", ctx.upload(x), ctx.upload(wt), ctx.upload(b), y, n, c_in, h, w, c_out, pad, layout, relu)
    close(ctx.download(y, ref_d.shape), ref_d, atol=1e-5)
    # no bias
    y2 = ctx.empty(n * c_out * ho * wo)
    ctx.call("th_conv3x3_fwd", ctx.upload(x), ctx.upload(wt), None, y2, n, c_in, h, w, c_out, pad, layout, 0)
    ref2 = xt.conv2d(wtt, None, (1, 1), (pad, pad), (1, 1)) if layout == 0 else xt.conv2d_direct_3x3(wtt, None, (1, 1), (pad, pad))
    ref2_d = ref2.data()[:, :, :ho, :wo]
    close(ctx.download(y2, ref2_d.shape), ref2_d, atol=1e-5)


@pytest.mark.parametrize("n,c_in,h,w,c_out", [(2, 1, 6, 6, 4), (3, 8, 7, 7, 16), (2, 5, 4, 9, 3)])
def test_conv1x1_taper_layout(ctx, O, n, c_in, h, w, c_out):
    """Q4: the reference's 1x1 path reinterprets the NCHW buffer as [N*H*W, C]"""
    rng = np.random.default_rng(n + c_in + c_out)
    x = rng.uniform(-1, 1, (n, c_in, h, w)).astype(np.float32)
    wt = rng.uniform(-0.5, 0.5, (c_out, c_in, 1, 1)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, c_out).astype(np.float32)
    ref = O.Tensor(x).conv2d(O.Tensor(wt), O.Tensor(b), (1, 1), (0, 0), (1, 1))
    y = ctx.empty(n * c_out * h * w)
    ctx.call("th_conv1x1_fwd", ctx.upload(x), ctx.upload(wt), ctx.upload(b), y, n, c_in, h, w, c_out, 0, 0)
    close(ctx.download(y, ref.shape()), ref.data(), atol=1e-5)
    # standard layout = a true per-pixel channel mix
    ctx.call("th_conv1x1_fwd", ctx.upload(x), ctx.upload(wt), ctx.upload(b), y, n, c_in, h, w, c_out, 1, 0)
    std = np.einsum("oc,nchw->nohw", wt[:, :, 0, 0].astype(np.float64), x.astype(np.float64)) + b[None, :, None, None]
    close(ctx.download(y, std.shape), std, atol=1e-5)


@pytest.mark.parametrize("n,c_in,h,w,c_out", [(3, 4, 8, 8, 5), (2, 32, 14, 14, 32), (4, 1, 28, 28, 8), (2, 16, 7, 7, 24),
                                              # >= 2048 output pixels and C_in >= 8: the matrix-core weight-gradient kernel
                                              (12, 16, 14, 14, 24), (4, 40, 28, 28, 70), (50, 9, 7, 7, 16), (3, 32, 28, 28, 32),
                                              (16, 1, 28, 28, 32), (11, 3, 30, 30, 5),    # few channel pairs, many pixels: image slabs
                                              (40, 1, 28, 28, 32), (33, 1, 32, 32, 16), (32, 1, 9, 7, 40),    # one input channel, >= 32 images: an image per workgroup (conv1_wgrad_kernel)
                                              (130, 32, 14, 14, 64), (129, 64, 7, 7, 128), (128, 32, 28, 28, 32), (131, 64, 14, 14, 64)])   # >= 128 images: the input gradient as a one-stage chain (conv_layer_chain_kernel<.., LIN>)
@pytest.mark.parametrize("layout", [0, 1])
def test_conv3x3_bwd_full_mode(ctx, O, n, c_in, h, w, c_out, layout):
    """full_backward extension (not in the reference, Q2): checked against the oracle's
    differentiable im2col chain (layout 0) / the same chain on re-laid-out weights."""
    rng = np.random.default_rng(n * 3 + c_in + c_out)
    x = rng.uniform(-1, 1, (n, c_in, h, w)).astype(np.float32)
    wt = rng.uniform(-0.5, 0.5, (c_out, c_in, 3, 3)).astype(np.float32)
    gy = rng.uniform(-1, 1, (n, c_out, h, w)).astype(np.float32)
    k = c_in * 9
    # express a standard-layout filter in taper's layout so the oracle chain can differentiate it
    w_taper = wt if layout == 0 else np.ascontiguousarray(wt.reshape(c_out, k).T).reshape(c_out, c_in, 3, 3)
    O.Tape.reset()
    O.Tape.set_zero_sentinel(False)
    xt, wtt = O.Tensor(x).requires_grad(), O.Tensor(w_taper).requires_grad()
    out = xt.conv2d(wtt, None, (1, 1), (1, 1), (1, 1), mode=1)
    (out * O.Tensor(gy)).sum(None, False).backward()
    O.Tape.set_zero_sentinel(True)
    gx_ref = xt.grad()
    gw_ref = wtt.grad() if layout == 0 else np.ascontiguousarray(wtt.grad().reshape(k, c_out).T).reshape(c_out, c_in, 3, 3)
    # accumulate = 0 (the slot was None): whatever the buffers held is overwritten, no zero fill needed
    gx, gw = ctx.upload(np.full(x.size, 7.0, np.float32)), ctx.upload(np.full(wt.size, -3.0, np.float32))
    ctx.call("th_conv3x3_bwd_input", ctx.upload(gy), ctx.upload(wt), gx, n, c_in, h, w, c_out, 1, layout, 0)
    ctx.call("th_conv3x3_bwd_weight", ctx.upload(x), ctx.upload(gy), gw, n, c_in, h, w, c_out, 1, layout, 0)
    close(ctx.download(gx, x.shape), gx_ref, atol=1e-4)
    close(ctx.download(gw, wt.shape), gw_ref, atol=1e-4 + 2e-7 * n * h * w)   # a reordered sum of n*h*w terms of O(1)
    # accumulate = 1 (ops.rs:126-129 semantics of a slot that is Some): a second call doubles both
    ctx.call("th_conv3x3_bwd_weight", ctx.upload(x), ctx.upload(gy), gw, n, c_in, h, w, c_out, 1, layout, 1)
    close(ctx.download(gw, wt.shape), 2 * gw_ref, atol=2e-4 + 4e-7 * n * h * w)
    ctx.call("th_conv3x3_bwd_input", ctx.upload(gy), ctx.upload(wt), gx, n, c_in, h, w, c_out, 1, layout, 1)
    close(ctx.download(gx, x.shape), 2 * gx_ref, atol=2e-4)
    O.Tape.reset()


POOL_CASES = [  # n, c, h, w, k, s, pad
    (4, 32, 28, 28, (2, 2), (2, 2), (0, 0)), (3, 64, 14, 14, (2, 2), (2, 2), (0, 0)), (2, 3, 7, 7, (2, 2), None, (0, 0)),
    (2, 3, 9, 8, (3, 3), (2, 2), (1, 1)), (1, 2, 5, 5, (3, 3), (1, 1), (1, 1)), (2, 2, 6, 6, (2, 3), (1, 2), (0, 1)),
]


@pytest.mark.parametrize("n,c,h,w,k,s,pad", POOL_CASES)
def test_maxpool_bit_exact(ctx, O, n, c, h, w, k, s, pad):
    rng = np.random.default_rng(n + c + h + w)
    x = rng.integers(-3, 4, (n, c, h, w)).astype(np.float32)   # many ties: first max (kh outer, kw inner) must win
    x[0, 0, :2, :2] = np.nan                                   # NaN never wins (tensor.rs:1451)
    O.Tape.reset()
    O.Tape.set_zero_sentinel(False)
    xt = O.Tensor(x).requires_grad()
    ref, ref_idx = xt.max_pool2d(k, s, pad, zero_first=True, return_indices=True)
    _, _, ho, wo = ref.shape()
    gout = rng.uniform(-1, 1, (n, c, ho, wo)).astype(np.float32)
    xt.set_grad(np.full(x.shape, 5.0, np.float32))   # pre-existing grad is ZEROED by the reference backward (Q5)
    (ref * O.Tensor(gout)).sum(None, False).backward()
    O.Tape.set_zero_sentinel(True)
    sh, sw = s or (0, 0)
    y, am = ctx.empty(n * c * ho * wo), ctx.empty(n * c * ho * wo, np.int64)
    ctx.call("th_maxpool2d_fwd", ctx.upload(x), y, am, n, c, h, w, k[0], k[1], sh, sw, pad[0], pad[1])
    np.testing.assert_array_equal(ctx.download(am, (n, c, ho, wo), np.int64), ref_idx)   # absolute flat index, bit-exact
    np.testing.assert_array_equal(ctx.download(y, (n, c, ho, wo)), ref.data())
    gin = ctx.upload(np.full(x.shape, 5.0, np.float32))
    ctx.call("th_maxpool2d_bwd", ctx.upload(gout), am, gin, n, c, h, w, k[0], k[1], sh, sw, pad[0], pad[1], 1)
    np.testing.assert_array_equal(ctx.download(gin, x.shape), xt.grad())                 # same add order: bit-exact
    O.Tape.reset()


@pytest.mark.parametrize("n,c,h,w,k,s,pad", [(2, 3, 8, 8, (2, 2), (2, 2), (0, 0)), (1, 2, 9, 7, (3, 3), (2, 2), (1, 1)),
                                             (2, 2, 6, 6, (2, 2), (2, 2), (2, 2)), (1, 1, 5, 5, (1, 1), (1, 1), (1, 1))])
def test_maxpool_windows_without_a_maximum(ctx, O, n, c, h, w, k, s, pad):
    """A window whose entries are all NaN / -inf -- or all padding (pad >= kernel) -- keeps the reference's default index `in_base`
    (tensor.rs:1432): value -inf, index = pixel (0,0) of ITS plane, and the backward adds that window's gradient there
    (tensor.rs:1504-1514) wherever the window lies.  Bit-exact vs the oracle, forward indices and backward."""
    rng = np.random.default_rng(h * 10 + w)
    x = rng.integers(-3, 4, (n, c, h, w)).astype(np.float32)
    x[0, 0, h - 4:, w - 4:] = np.nan          # all-NaN windows far from pixel (0,0)
    x[n - 1, c - 1, 2:, :] = -np.inf          # -inf never beats the -inf start either
    x[n - 1, c - 1, 0, 0] = np.nan
    O.Tape.reset()
    O.Tape.set_zero_sentinel(False)
    xt = O.Tensor(x).requires_grad()
    ref, ref_idx = xt.max_pool2d(k, s, pad, zero_first=True, return_indices=True)
    _, _, ho, wo = ref.shape()
    gout = rng.uniform(-1, 1, (n, c, ho, wo)).astype(np.float32)
    (ref * O.Tensor(gout)).sum(None, False).backward()
    O.Tape.set_zero_sentinel(True)
    # the case is really exercised: some window away from the origin points at its plane's first pixel
    plane_base = (np.arange(n * c) * h * w).reshape(n, c, 1, 1)
    dflt = (ref_idx == plane_base) & np.isneginf(ref.data().reshape(n, c, ho, wo))
    assert dflt[:, :, 1:, :].any() or dflt[:, :, :, 1:].any()
    y, am = ctx.empty(n * c * ho * wo), ctx.empty(n * c * ho * wo, np.int64)
    ctx.call("th_maxpool2d_fwd", ctx.upload(x), y, am, n, c, h, w, k[0], k[1], s[0], s[1], pad[0], pad[1])
    np.testing.assert_array_equal(ctx.download(am, (n, c, ho, wo), np.int64), ref_idx)
    np.testing.assert_array_equal(ctx.download(y, (n, c, ho, wo)), ref.data().reshape(n, c, ho, wo))
    gin = ctx.upload(np.full(x.shape, 5.0, np.float32))
    ctx.call("th_maxpool2d_bwd", ctx.upload(gout), am, gin, n, c, h, w, k[0], k[1], s[0], s[1], pad[0], pad[1], 1)
    np.testing.assert_array_equal(ctx.download(gin, x.shape), xt.grad().reshape(x.shape))
    O.Tape.reset()


@pytest.mark.parametrize("n,c,h,w", [(4, 32, 28, 28), (3, 64, 14, 14), (2, 3, 8, 8), (1, 1, 2, 2), (2, 5, 30, 18), (1, 2, 40, 40)])
@pytest.mark.parametrize("zero_first", [1, 0])
def test_maxpool2_bwd_fast_path_equals_the_general_kernel(ctx, n, c, h, w, zero_first):
    """th_maxpool2d_bwd on 2x2 / stride 2 / unpadded pools takes a wave-per-plane kernel; TAPER_POOL_BWD_GENERAL is not set here, so the
    comparison is against the scatter itself: gin[argmax[o]] (+)= gout[o] in ascending o (tensor.rs:1496-1519), with windows that kept
    the default index (all NaN) landing on their plane's pixel (0,0) -- bit-exact, both with zero_first (Q5) and accumulating."""
    rng = np.random.default_rng(n * 100 + c * 10 + h)
    x = rng.integers(-3, 4, (n, c, h, w)).astype(np.float32)
    if h >= 8:
        x[0, 0, h - 4:, w - 4:] = np.nan      # default-index windows far from the origin
        x[n - 1, c - 1, :2, :2] = np.nan      # ... and at it
    ho, wo = h // 2, w // 2
    y, am = ctx.empty(n * c * ho * wo), ctx.empty(n * c * ho * wo, np.int64)
    ctx.call("th_maxpool2d_fwd", ctx.upload(x), y, am, n, c, h, w, 2, 2, 2, 2, 0, 0)
    idx = ctx.download(am, (n * c * ho * wo,), np.int64)
    gout = rng.uniform(-1, 1, n * c * ho * wo).astype(np.float32)
    old = rng.uniform(-1, 1, x.size).astype(np.float32)
    ref = np.zeros(x.size, np.float32) if zero_first else old.copy()
    for o in range(idx.size):                  # sequential, ascending: the reference's order
        ref[idx[o]] = np.float32(ref[idx[o]] + gout[o])
    gin = ctx.upload(old)
    ctx.call("th_maxpool2d_bwd", ctx.upload(gout), am, gin, n, c, h, w, 2, 2, 2, 2, 0, 0, zero_first)
    np.testing.assert_array_equal(ctx.download(gin, (x.size,)), ref)


@pytest.mark.parametrize("n,c,h,w", [(4, 32, 28, 28), (3, 64, 14, 14), (2, 3, 8, 8), (1, 1, 2, 2), (2, 5, 30, 18)])
def test_maxpool2d_relu_bwd_equals_pool_bwd_then_relu_bwd(ctx, n, c, h, w):
    """th_maxpool2d_relu_bwd == th_maxpool2d_bwd(zero_first) followed by th_relu_bwd on the pool's input, bit for bit -- including planes
    whose default-index windows send their gradient to pixel (0,0), where THAT pixel's output decides the mask."""
    from taper_amd import hip
    assert hip.hip.th_maxpool2d_relu_bwd_supported(n, c, h, w, 2, 2, 2, 2, 0, 0) == 1
    assert hip.hip.th_maxpool2d_relu_bwd_supported(n, c, h, w, 3, 3, 2, 2, 1, 1) == 0
    rng = np.random.default_rng(n * 100 + c * 10 + h)
    x = np.maximum(rng.integers(-3, 4, (n, c, h, w)).astype(np.float32), 0)     # a ReLU's output: many zeros, many ties
    if h >= 8:
        x[0, 0, h - 4:, w - 4:] = np.nan
        x[0, 0, 0, 0] = 2.0                   # pixel (0,0) > 0: the default-index windows' gradients survive the mask
        x[n - 1, c - 1, h - 2:, w - 2:] = np.nan
        x[n - 1, c - 1, 0, 0] = 0.0           # ... and here they do not
    ho, wo = h // 2, w // 2
    dx = ctx.upload(x)
    y, am = ctx.empty(n * c * ho * wo), ctx.empty(n * c * ho * wo, np.int64)
    ctx.call("th_maxpool2d_fwd", dx, y, am, n, c, h, w, 2, 2, 2, 2, 0, 0)
    gout = ctx.upload(rng.uniform(-1, 1, n * c * ho * wo).astype(np.float32))
    ref, tmp = ctx.empty(x.size), ctx.upload(np.full(x.size, 9.0, np.float32))
    ctx.call("th_maxpool2d_bwd", gout, am, tmp, n, c, h, w, 2, 2, 2, 2, 0, 0, 1)
    ctx.call("th_relu_bwd", dx, tmp, ref, x.size, 0)
    got, sums = ctx.upload(np.full(x.size, -5.0, np.float32)), ctx.upload(np.full(n * c, 3.0, np.float32))
    ctx.call("th_maxpool2d_relu_bwd", gout, am, y, dx, got, sums, n, c, h, w)
    want = ctx.download(ref, (x.size,))
    np.testing.assert_array_equal(ctx.download(got, (x.size,)), want)
    # the plane sums of the result (the rows of the producer's bias gradient), and the same call without them
    close(ctx.download(sums, (n * c,)), want.reshape(n * c, h * w).astype(np.float64).sum(1), atol=1e-5 * np.sqrt(h * w))
    got2 = ctx.empty(x.size)
    ctx.call("th_maxpool2d_relu_bwd", gout, am, y, dx, got2, None, n, c, h, w)
    np.testing.assert_array_equal(ctx.download(got2, (x.size,)), want)


@pytest.mark.parametrize("n,c,hw", [(4, 32, 784), (3, 64, 196), (5, 128, 49), (2, 3, 30), (1, 1, 1), (7, 5, 4096)])
def test_relu_bwd_plane_sums_and_bias_from_them(ctx, n, c, hw):
    """th_relu_bwd_plane_sums == th_relu_bwd bit for bit, + the per-plane sums; th_bias_grad_plane_sums adds those over the images:
    together th_relu_bwd + th_bias_grad_nchw (tensor.rs:2017-2024) in one pass over the map."""
    rng = np.random.default_rng(n * 7 + c + hw)
    y = np.maximum(rng.uniform(-1, 1, (n, c, hw)), 0).astype(np.float32)
    g = rng.uniform(-1, 1, (n, c, hw)).astype(np.float32)
    dy, dg = ctx.upload(y), ctx.upload(g)
    ref, got, sums = ctx.empty(y.size), ctx.upload(np.full(y.size, 4.0, np.float32)), ctx.empty(n * c)
    ctx.call("th_relu_bwd", dy, dg, ref, y.size, 0)
    ctx.call("th_relu_bwd_plane_sums", dy, dg, got, sums, n, c, hw)
    want = ctx.download(ref, (n, c, hw))
    np.testing.assert_array_equal(ctx.download(got, (n, c, hw)), want)
    np.testing.assert_array_equal(want, np.where(y > 0, g, 0).astype(np.float32))
    close(ctx.download(sums, (n, c)), want.astype(np.float64).sum(2), atol=2e-6 * hw)
    gb = ctx.upload(np.full(c, 2.0, np.float32))
    ctx.call("th_bias_grad_plane_sums", sums, gb, n, c, 0)
    close(ctx.download(gb, (c,)), want.astype(np.float64).sum((0, 2)), atol=2e-6 * hw * n)
    ctx.call("th_bias_grad_plane_sums", sums, gb, n, c, 1)
    close(ctx.download(gb, (c,)), 2 * want.astype(np.float64).sum((0, 2)), atol=4e-6 * hw * n)


@pytest.mark.parametrize("n,c,hw", [(5, 128, 49), (3, 7, 196), (2, 2, 1), (1, 3, 100)])
def test_avgpool_global_relu_bwd_equals_pool_bwd_then_relu_bwd(ctx, n, c, hw):
    """th_avgpool2d_global_relu_bwd == th_avgpool2d_bwd into a zeroed slot followed by th_relu_bwd, + the plane sums."""
    rng = np.random.default_rng(n + c + hw)
    y = np.maximum(rng.uniform(-1, 1, (n, c, hw)), 0).astype(np.float32)
    g = rng.uniform(-1, 1, (n, c)).astype(np.float32)
    dy, dg = ctx.upload(y), ctx.upload(g)
    tmp, ref = ctx.zeros(y.size), ctx.empty(y.size)
    ctx.call("th_avgpool2d_bwd", dg, tmp, n, c, hw, 1, hw, 1, hw, 1, 0, 0)
    ctx.call("th_relu_bwd", dy, tmp, ref, y.size, 0)
    got, sums = ctx.upload(np.full(y.size, 3.0, np.float32)), ctx.empty(n * c)
    ctx.call("th_avgpool2d_global_relu_bwd", dg, dy, got, sums, n, c, hw)
    want = ctx.download(ref, (n, c, hw))
    np.testing.assert_array_equal(ctx.download(got, (n, c, hw)), want)
    close(ctx.download(sums, (n, c)), want.astype(np.float64).sum(2), atol=2e-6 * hw)


@pytest.mark.parametrize("n,c,h,w,k,s,pad", POOL_CASES + [(5, 128, 7, 7, (7, 7), (7, 7), (0, 0)), (2, 4, 7, 7, (7, 7), (1, 1), (0, 0))])
def test_avgpool(ctx, O, n, c, h, w, k, s, pad):
    rng = np.random.default_rng(n * c + h)
    x = rng.uniform(-1, 1, (n, c, h, w)).astype(np.float32)
    O.Tape.reset()
    O.Tape.set_zero_sentinel(False)
    xt = O.Tensor(x).requires_grad()
    ref = xt.avg_pool2d(k, s, pad)
    _, _, ho, wo = ref.shape()
    gout = rng.uniform(-1, 1, (n, c, ho, wo)).astype(np.float32)
    (ref * O.Tensor(gout)).sum(None, False).backward()
    O.Tape.set_zero_sentinel(True)
    sh, sw = s or (0, 0)
    y = ctx.empty(n * c * ho * wo)
    ctx.call("th_avgpool2d_fwd", ctx.upload(x), y, n, c, h, w, k[0], k[1], sh, sw, pad[0], pad[1])
    close(ctx.download(y, (n, c, ho, wo)), ref.data(), atol=1e-6)
    gin = ctx.zeros(x.size)
    ctx.call("th_avgpool2d_bwd", ctx.upload(gout), gin, n, c, h, w, k[0], k[1], sh, sw, pad[0], pad[1])
    close(ctx.download(gin, x.shape), xt.grad(), atol=1e-7)
    O.Tape.reset()


def test_bias_nchw(ctx):
    rng = np.random.default_rng(5)
    n, c, hw = 6, 20, 49
    x = rng.uniform(-1, 1, (n, c, hw)).astype(np.float32)
    b = rng.uniform(-1, 1, c).astype(np.float32)
    y = ctx.empty(x.size)
    ctx.call("th_bias_add_nchw", ctx.upload(x), ctx.upload(b), y, n, c, hw, 1)
    np.testing.assert_array_equal(ctx.download(y, x.shape), np.maximum(x + b[None, :, None], 0))
    gb0 = rng.uniform(-1, 1, c).astype(np.float32)
    gb = ctx.upload(gb0)
    ctx.call("th_bias_grad_nchw", ctx.upload(x), gb, n, c, hw)
    np.testing.assert_allclose(ctx.download(gb, c), gb0 + x.astype(np.float64).sum((0, 2)), rtol=RTOL, atol=1e-5)


# ------------------------------------------------------------------ optimizers / data
@pytest.mark.parametrize("padded", [True, False])
def test_adam_matches_oracle_over_steps(ctx, O, padded):
    """padded: every tensor starts on a 16-byte boundary (the host arena's layout: all dwordx4 quads);
    tightly packed: quads straddle tensors, one of them grad-less (element-wise path)"""
    rng = np.random.default_rng(9)
    sizes = [128 * 784, 128, 1280, 10, 7] if padded else [128 * 784 + 1, 7, 127, 1281, 10, 7, 3]   # one without a gradient (Q8)
    has = [1, 1, 1, 1, 0] if padded else [1, 0, 1, 1, 1, 0, 1]
    offs = np.zeros(len(sizes) + 1, np.int64)
    for i, s in enumerate(sizes):
        offs[i + 1] = offs[i] + ((s + 3) // 4 * 4 if padded else s)
    total = int(offs[-1])
    p0 = rng.uniform(-0.1, 0.1, total).astype(np.float32)
    params = [O.Tensor(p0[offs[i]:offs[i] + s]).requires_grad() for i, s in enumerate(sizes)]
    oopt = O.Adam(params, 1e-3, None, None, 1e-4)
    dp, dm, dv = ctx.upload(p0), ctx.zeros(total), ctx.zeros(total)
    doffs, dhas = ctx.upload(offs), ctx.upload(np.array(has, np.int32))
    state = ctx.upload(np.zeros(2, np.int32))
    lr = ctx.upload(np.array([1e-3], np.float32))
    for step in range(25):
        g = (rng.standard_normal(total) * 0.01).astype(np.float32)
        for i, s in enumerate(sizes):
            params[i].set_grad(g[offs[i]:offs[i] + s] if has[i] else None)
        oopt.step()
        dg = ctx.upload(g)
        ctx.call("th_adam_step", dp, dg, dm, dv, doffs, dhas, len(sizes), total, state, lr, 0.9, 0.999, 1e-8, 1e-4, 0)
    got = ctx.download(dp, total)
    assert ctx.download(state, 2, np.int32)[0] == 25 == oopt.t()
    for i, s in enumerate(sizes):
        np.testing.assert_allclose(got[offs[i]:offs[i] + s], params[i].data(), rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(ctx.download(dm, total)[offs[i]:offs[i] + s], oopt.m(i), rtol=RTOL, atol=1e-9)
    for i, s in enumerate(sizes):
        if not has[i]:
            np.testing.assert_array_equal(got[offs[i]:offs[i] + s], p0[offs[i]:offs[i] + s])   # grad-less: untouched


def test_powi_matches(O):
    def powisf2(a, b):  # compiler-rt __powisf2 in float32, what f32::powi lowers to (optim.rs:87-88)
        a, r = np.float32(a), np.float32(1)
        while True:
            if b & 1:
                r = np.float32(r * a)
            b //= 2
            if b == 0:
                return r
            a = np.float32(a * a)
    for b, t in [(0.9, 1), (0.9, 7), (0.999, 1000), (0.999, 12345), (0.5, 31)]:
        assert O.powi(b, t) == powisf2(b, t)
        assert O.powi(b, t) == pytest.approx(float(np.float32(b)) ** t, rel=2e-3)


def test_sgd(ctx):
    p0 = np.linspace(-1, 1, 16).astype(np.float32)
    g = np.linspace(1, 2, 16).astype(np.float32)
    dp = ctx.upload(p0)
    ctx.call("th_sgd_step", dp, ctx.upload(g), ctx.upload(np.array([0, 8, 16], np.int64)), ctx.upload(np.array([1, 0], np.int32)), 2, 16,
             ctx.upload(np.array([0.1], np.float32)))
    ref = p0.copy()
    ref[:8] -= np.float32(0.1) * g[:8]
    np.testing.assert_array_equal(ctx.download(dp, 16), ref)


def test_gather_batch_and_u8(ctx, O):
    rng = np.random.default_rng(11)
    n = 300
    px = rng.integers(0, 256, (n, 784)).astype(np.uint8)
    labels = rng.integers(0, 10, n).astype(np.float32)
    imgs = ctx.empty(n * 784)
    ctx.call("th_u8_to_unit_f32", ctx.upload(px), imgs, n * 784)
    ref_imgs = px.astype(np.float32) / np.float32(255.0)          # data/mnist.rs:226
    np.testing.assert_array_equal(ctx.download(imgs, (n, 784)), ref_imgs)
    idx = rng.permutation(n).astype(np.int32)
    xb, yb = ctx.empty(64 * 784), ctx.empty(64)
    cursor = ctx.upload(np.array([100], np.int64))
    ctx.call("th_gather_batch", imgs, ctx.upload(labels), ctx.upload(idx), n, cursor, 64, 784, xb, yb)
    rx, ry = O.get_batch(ref_imgs, labels, idx[100:164])
    np.testing.assert_array_equal(ctx.download(xb, (64, 784)), rx)
    np.testing.assert_array_equal(ctx.download(yb, 64), ry)


def test_log_step_and_graph_replay(ctx):
    """device-side step log + cursor under hipGraph replay"""
    state = ctx.upload(np.zeros(2, np.int64))
    metrics = ctx.zeros(2 * 8)
    loss, nc = ctx.upload(np.array([1.5], np.float32)), ctx.upload(np.array([3.0], np.float32))
    ctx.graph_begin()
    ctx.call("th_axpy", 1.0, nc, loss, 1)                       # loss += 3 every replay
    ctx.call("th_log_step", loss, nc, metrics, 8, state, 64)
    g = ctx.graph_end()
    for _ in range(5):
        ctx.graph_launch(g)
    ctx.sync()
    ctx.graph_destroy(g)
    np.testing.assert_array_equal(ctx.download(state, 2, np.int64), [5, 320])
    m = ctx.download(metrics, (8, 2))
    np.testing.assert_array_equal(m[:5, 0], [4.5, 7.5, 10.5, 13.5, 16.5])


def test_error_paths(ctx):
    """the reference panics; the C ABI returns an error code + message"""
    from taper_amd._lib import TaperError
    with pytest.raises(TaperError):
        ctx.call("th_sgemm", 0, 0, -1, 2, 2, 1.0, None, None, 0.0, None)
    with pytest.raises(TaperError):
        ctx.call("th_conv3x3_fwd", None, None, None, None, 1, 1, 4, 4, 1, 1, 0, 0)
    with pytest.raises(TaperError):
        x = ctx.zeros(16)
        ctx.call("th_conv3x3_fwd", x, x, None, x, 1, 1, 4, 4, 1, 2, 0, 0)   # pad 2 unsupported


def test_rccl_single_rank_allreduce(ctx):
    """th_comm_* over RCCL with one rank: sum over ranks is the identity, then the 1/W scale"""
    import ctypes
    from taper_amd._lib import hip as lib, th_check
    uid = (ctypes.c_uint8 * 128)()
    th_check(lib.th_comm_unique_id(uid), "th_comm_unique_id")
    comm = ctypes.c_void_p()
    th_check(lib.th_comm_init_rank(ctx.h, 1, 0, uid, ctypes.byref(comm)), "th_comm_init_rank")
    x = np.arange(101_772, dtype=np.float32)
    d = ctx.upload(x)
    th_check(lib.th_allreduce_sum_scale(comm, ctx.h, int(d), x.size, 0.5), "th_allreduce_sum_scale")
    np.testing.assert_array_equal(ctx.download(d, x.size), x * np.float32(0.5))
    # scale == 1/n_ranks: the mean runs as ncclAvg inside the collective (no scale launch)
    th_check(lib.th_allreduce_sum_scale(comm, ctx.h, int(d), x.size, 1.0), "th_allreduce_sum_scale")
    np.testing.assert_array_equal(ctx.download(d, x.size), x * np.float32(0.5))
    lib.th_comm_destroy(comm)


@pytest.mark.parametrize("n,c,h,w", [(256, 64, 7, 7), (6, 5, 3, 3), (40, 128, 7, 7), (9, 16, 10, 10), (300, 32, 14, 14)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_pooled_bias_gradients(ctx, n, c, h, w, accumulate):
    """bias gradient of a Conv2dReLU straight from the gradient of the pool behind it:
    max pool (th_bias_grad_nchw_masked on the pooled tensors) and global average pool (th_bias_grad_avgpool_masked)"""
    rng = np.random.default_rng(n + c + h)
    hw = h * w
    dyp = rng.standard_normal((n, c, hw)).astype(np.float32)
    yp = np.maximum(rng.standard_normal((n, c, hw)), 0).astype(np.float32)       # pooled ReLU outputs (zeros included)
    db0 = rng.standard_normal(c).astype(np.float32)
    out = ctx.upload(db0)
    ctx.call("th_bias_grad_nchw_masked", ctx.upload(dyp), ctx.upload(yp), out, n, c, hw, accumulate)
    ref = (dyp.astype(np.float64) * (yp > 0)).sum((0, 2)) + (db0 if accumulate else 0)
    np.testing.assert_allclose(ctx.download(out, c), ref, rtol=1e-4, atol=1e-4 * float(np.abs(ref).max()) + 1e-6)
    g = rng.standard_normal((n, c)).astype(np.float32)                            # gradient of the global average pool's output
    out2 = ctx.upload(db0)
    ctx.call("th_bias_grad_avgpool_masked", ctx.upload(g), ctx.upload(yp), out2, n, c, hw, accumulate)
    ref2 = (g.astype(np.float64)[:, :, None] / hw * (yp > 0)).sum((0, 2)) + (db0 if accumulate else 0)
    np.testing.assert_allclose(ctx.download(out2, c), ref2, rtol=1e-4, atol=1e-4 * float(np.abs(ref2).max()) + 1e-6)


@pytest.mark.parametrize("n,ci,h,w,co,pad", [(4, 1, 28, 28, 32, 1), (5, 32, 28, 28, 32, 1), (3, 32, 14, 14, 64, 1), (7, 64, 14, 14, 64, 1),
                                             (2, 16, 8, 6, 8, 1), (3, 8, 10, 12, 20, 0), (9, 8, 4, 4, 12, 1), (1, 24, 30, 62, 36, 1)])
def test_conv3x3_pool2_fused(ctx, n, ci, h, w, co, pad):
    """conv + bias + ReLU + 2x2 max pool in one launch == th_conv3x3_fwd followed by th_maxpool2d_fwd, bit for bit
    (same k-ordered products per pixel, same maxima); only the pooled tensor is written"""
    rng = np.random.default_rng(n * 31 + ci + h + w + co)
    x = rng.standard_normal((n, ci, h, w)).astype(np.float32)
    wt = (rng.standard_normal((co, ci, 3, 3)) * 0.2).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    xd, wd, bd = ctx.upload(x), ctx.upload(wt), ctx.upload(b)
    full, pooled_ref, arg = ctx.empty(n * co * ho * wo), ctx.empty(n * co * (ho // 2) * (wo // 2)), ctx.empty(n * co * (ho // 2) * (wo // 2), np.int64)
    ctx.call("th_conv3x3_fwd", xd, wd, bd, full, n, ci, h, w, co, pad, 0, 1)
    ctx.call("th_maxpool2d_fwd", full, pooled_ref, arg, n, co, ho, wo, 2, 2, 2, 2, 0, 0)
    fused = ctx.empty(n * co * (ho // 2) * (wo // 2))
    ctx.call("th_conv3x3_pool2_fwd", xd, wd, bd, fused, n, ci, h, w, co, pad, 1)
    np.testing.assert_array_equal(ctx.download(fused, (n, co, ho // 2, wo // 2)), ctx.download(pooled_ref, (n, co, ho // 2, wo // 2)))


def test_conv3x3_pool2_limits(ctx):
    from taper_amd._lib import TaperError
    z = ctx.zeros(4096)
    with pytest.raises(TaperError, match="even output"):
        ctx.call("th_conv3x3_pool2_fwd", z, z, None, z, 1, 8, 5, 6, 8, 1, 1)      # odd height
    with pytest.raises(TaperError, match="even output"):
        ctx.call("th_conv3x3_pool2_fwd", z, z, None, z, 1, 4, 6, 6, 8, 1, 1)      # c_in = 4: not the matrix-core path


# ------------------------------------------------------------------ sum(dim) / max(dim) on any rank (tensor.rs:890-1018, 1021-1071)
@pytest.mark.parametrize("shape", [(5, 7), (3, 4, 5), (2, 3, 4, 5), (1, 6, 1), (4, 1, 3), (2, 1), (1, 1, 3), (7,)])
def test_sum_and_max_over_any_dimension_match_the_reference_literally(shape):
    """the host mirror accepts every (shape, dim) the reference accepts and gives what it gives -- the oracle restates tensor.rs:917-937 /
    960-994 / 1042-1066 element by element, Q14's colliding max indices on rank > 2 and the backward's skipped coordinate on tiny outputs
    included: values bit-exact (a row's terms are added in input order), indices exact, the gradient of sum exact"""
    from tests import backends
    H, Orc = backends.get("hip"), backends.get("oracle")
    rng = np.random.default_rng(sum(shape) + len(shape))
    x = rng.standard_normal(shape).astype(np.float32)
    x.reshape(-1)[::3] = np.float32(0.5)                   # ties: the first of equal maxima wins
    if x.size > 4:
        x.reshape(-1)[1] = np.nan                           # a NaN never wins (tensor.rs:1062: strict >)
        x.reshape(-1)[-1] = -np.inf
    xs = np.nan_to_num(x, nan=0.25, neginf=-3.0)            # (sum: finite data)
    up = rng.standard_normal(shape).astype(np.float32)
    for dim in range(len(shape)):
        hv, hi = H.Tensor(x).max(dim)
        ov, oi = Orc.Tensor(x).max(dim)
        np.testing.assert_array_equal(hv.data(), ov.data())
        np.testing.assert_array_equal(hi.data(), oi.data())
        for keepdim in (False, True):
            grads = []
            for B in (H, Orc):
                B.Tape.reset()
                t = B.Tensor(xs).requires_grad()
                s = t.sum(dim, keepdim)
                w = B.Tensor(up.sum(axis=dim, keepdims=keepdim).reshape(np.shape(s.data())).astype(np.float32))
                (s * w).sum(None, False).backward()
                grads.append((np.asarray(s.data()), np.asarray(t.grad())))
                B.Tape.reset()
            outer, inner = int(np.prod(shape[:dim], dtype=np.int64)), int(np.prod(shape[dim + 1:], dtype=np.int64))
            if (inner != 1 and outer != 1) or grads[1][0].size < len(shape):      # the literal kernel: a row's terms in input order
                np.testing.assert_array_equal(grads[0][0], grads[1][0], err_msg=f"sum({dim}, {keepdim}) of {shape}")
            else:                                                                   # first / last dimension: the hot path's wave-shuffle sums
                np.testing.assert_allclose(grads[0][0], grads[1][0], rtol=1e-6, atol=1e-6, err_msg=f"sum({dim}, {keepdim}) of {shape}")
            np.testing.assert_array_equal(grads[0][1].reshape(shape), grads[1][1].reshape(shape), err_msg=f"grad of sum({dim}, {keepdim}) of {shape}")
