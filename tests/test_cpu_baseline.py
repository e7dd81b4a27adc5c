"""The cpu_baseline build of the oracle sources (`make -C oracle fast`: matrixmultiply-style packed sgemm on one
thread + the reference's rayon loops as OpenMP loops + optional vendor cblas_sgemm; what bench.py times beside the GPU)
must compute what the plain-loop parity oracle computes: the packed kernel on every layout / ragged shape / alpha-beta
combination of `sgemm_rowmajor` (/root/reference/src/gemm.rs:72-119), and whole training steps of the MLP and the CNN."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
from tests import backends

_f32p = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def fast(tmp_path_factory):
    so = O.build_fast(str(tmp_path_factory.mktemp("oracle_fast")))
    lib = C.CDLL(str(so))
    lib.ot_baseline_flavour.restype = C.c_int
    assert lib.ot_baseline_flavour() == 3          # packed sgemm + OpenMP loops
    assert O.lib.ot_baseline_flavour() == 0        # the parity oracle stays the plain k-ordered loop, one thread
    return so, lib


def _sgemm(lib, ta, tb, m, n, k, alpha, a, b, beta, c):
    lib.ot_sgemm_rowmajor(ta, tb, m, n, k, C.c_float(alpha), a.ctypes.data_as(_f32p), b.ctypes.data_as(_f32p), C.c_float(beta),
                          c.ctypes.data_as(_f32p))


SHAPES = [(1, 1, 1), (7, 5, 3), (2, 3, 2), (64, 128, 784), (64, 10, 128), (784, 128, 64), (128, 10, 64), (64, 128, 10), (13, 33, 300),
          (100, 70, 513), (257, 129, 31), (145, 4100, 17)]


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_packed_sgemm_equals_plain_loop(fast, m, n, k):
    _, lib = fast
    rng = np.random.default_rng(m * 7 + n * 3 + k)
    for ta in (0, 1):
        for tb in (0, 1):
            for alpha, beta in ((1.0, 0.0), (0.5, 1.0), (2.0, -0.5), (0.0, 2.0)):
                a = rng.standard_normal(m * k).astype(np.float32)
                b = rng.standard_normal(k * n).astype(np.float32)
                c0 = rng.standard_normal(m * n).astype(np.float32)
                c1, c2 = c0.copy(), c0.copy()
                _sgemm(lib, ta, tb, m, n, k, alpha, a, b, beta, c1)
                _sgemm(O.lib, ta, tb, m, n, k, alpha, a, b, beta, c2)
                np.testing.assert_allclose(c1, c2, rtol=1e-4, atol=1e-4 * max(1.0, float(np.abs(c2).max())))


def test_smoke_rs_gemm_kat_on_the_packed_kernel(fast):
    """tests/smoke.rs:46-70: [[1,2,3],[4,5,6]] x [[7,8],[9,10],[11,12]] = [[58,64],[139,154]]"""
    _, lib = fast
    a = np.arange(1, 7, dtype=np.float32)
    b = np.arange(7, 13, dtype=np.float32)
    c = np.zeros(4, np.float32)
    _sgemm(lib, 0, 0, 2, 2, 3, 1.0, a, b, 0.0, c)
    assert c.tolist() == [58.0, 64.0, 139.0, 154.0]


def test_vendor_cblas_leg_matches(fast):
    """`--features blas` analogue (gemm.rs:21-47): RowMajor, lda = m|k, ldb = k|n, ldc = n through the box's cblas_sgemm"""
    found = O.find_cblas()
    if found is None:
        pytest.skip("no CBLAS on this box")
    _, lib = fast
    addr, ilp64, _, _keep = found
    lib.ot_baseline_set_cblas.argtypes = [C.c_void_p, C.c_int]
    lib.ot_baseline_set_cblas(addr, ilp64)
    try:
        rng = np.random.default_rng(0)
        for (m, n, k) in [(64, 128, 784), (13, 33, 300), (784, 128, 64)]:
            for ta in (0, 1):
                for tb in (0, 1):
                    a, b = rng.standard_normal(m * k).astype(np.float32), rng.standard_normal(k * n).astype(np.float32)
                    c1 = rng.standard_normal(m * n).astype(np.float32)
                    c2 = c1.copy()
                    _sgemm(lib, ta, tb, m, n, k, 0.5, a, b, 1.0, c1)
                    _sgemm(O.lib, ta, tb, m, n, k, 0.5, a, b, 1.0, c2)
                    np.testing.assert_allclose(c1, c2, rtol=1e-4, atol=1e-4 * float(np.abs(c2).max()))
    finally:
        lib.ot_baseline_set_cblas(None, 0)


@pytest.mark.parametrize("key,batch,shape", [("mlp_baseline", 64, (64, 784)), ("cnn_simple", 16, (16, 1, 28, 28)),
                                             ("cnn_reference", 8, (8, 1, 28, 28))])
def test_baseline_build_trains_like_the_oracle(fast, key, batch, shape):
    """3 steps of {get_batch -> forward -> loss -> backward -> Adam} in the baseline build (4 OpenMP threads) against the
    same steps in the parity oracle: same weights afterwards"""
    so, _ = fast
    rng = np.random.default_rng(4)
    spec = backends.nonzero_biases(getattr(backends, key)(rng), rng)
    x, y = backends.mnist_like(rng, 3 * batch)
    finals = []
    try:
        for lib_path in (None, so):
            if lib_path is not None:
                O.use_library(lib_path)
                O.lib.ot_baseline_set_threads(4)
            ob = backends.get("oracle")
            ob.set_zero_sentinel(True)
            model = ob.sequential(spec)
            opt = O.Adam(model.parameters(), 1e-2, None, None, 1e-4)
            if lib_path is None:
                for s in range(3):
                    model.train_step(opt, x[s * batch:(s + 1) * batch], y[s * batch:(s + 1) * batch], shape)
            else:
                model.run_steps(opt, x, y, shape, 3)
            finals.append([p.data().copy() for p in model.parameters()])
    finally:
        O.use_library(O.build())
    for i, (a, b) in enumerate(zip(*finals)):
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=1e-2 * 2e-2, err_msg=f"param {i}")
