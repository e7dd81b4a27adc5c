"""The convolutional fronts of the two CNNs as ONE launch (th_conv_chain_fwd: a workgroup carries an image through every stage in
LDS) against the oracle's layer-by-layer tape ops (`conv2d_relu`, `max_pool2d`, global `avg_pool2d`: /root/reference/src/tensor.rs:1221-1285,
1391-1470, 1524-1660; the model rows of examples/train_mnist_cnn.rs:35-100) and, at batch 256, bit for bit against the layer-by-layer
HIP launches it replaces."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-4

REFERENCE = [(1, 32, 0), (32, 32, 1), (32, 64, 0), (64, 64, 1), (64, 128, 2)]      # (c_in, c_out, post): 0 none, 1 max-pool 2x2, 2 global mean
SIMPLE = [(1, 32, 1), (32, 64, 1)]


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


def _params(spec, seed):
    rng = np.random.default_rng(seed)
    out = []
    for c_in, c_out, _ in spec:
        bound = np.sqrt(6.0 / (c_in * 9))                                   # nn.rs:219-222
        out.append((rng.uniform(-bound, bound, (c_out, c_in, 3, 3)).astype(np.float32), rng.uniform(-0.1, 0.1, c_out).astype(np.float32)))
    return out


def _oracle_chain(O, x, spec, params):
    t = O.Tensor(x)
    cnt = None
    for (c_in, c_out, post), (w, b) in zip(spec, params):
        t = t.conv2d_relu(O.Tensor(w), O.Tensor(b), (1, 1), (1, 1), (1, 1))
        if post == 1:
            t = t.max_pool2d((2, 2), (2, 2), (0, 0))
        elif post == 2:
            d = t.data()
            cnt = (d > 0).sum(axis=(2, 3)).astype(np.float32)
            t = t.avg_pool2d((d.shape[2], d.shape[3]), (d.shape[2], d.shape[3]), (0, 0))
    return t.data(), cnt


def _hip_layered(ctx, x, spec, params):
    """the launches the chain replaces: th_conv3x3_fwd / th_conv3x3_pool2_fwd / th_conv3x3_gap_fwd, layer by layer"""
    n, hw = x.shape[0], x.shape[2]
    cur, cnt = ctx.upload(x), None
    for (c_in, c_out, post), (w, b) in zip(spec, params):
        dw, db = ctx.upload(w), ctx.upload(b)
        if post == 0:
            y = ctx.empty(n * c_out * hw * hw)
            ctx.call("th_conv3x3_fwd", cur, dw, db, y, n, c_in, hw, hw, c_out, 1, 0, 1)
        elif post == 1:
            y = ctx.empty(n * c_out * (hw // 2) ** 2)
            ctx.call("th_conv3x3_pool2_fwd", cur, dw, db, y, n, c_in, hw, hw, c_out, 1, 1)
            hw //= 2
        else:
            y, cnt = ctx.empty(n * c_out), ctx.empty(n * c_out)
            ctx.call("th_conv3x3_gap_fwd", cur, dw, db, y, cnt, n, c_in, hw, hw, c_out, 1, 1)
            hw = 1
        cur = y
    return cur, cnt, hw


def _hip_chain(ctx, x, spec, params, want_cnt=True, want_kind=None):
    from taper_amd import hip
    n, c0, hw = x.shape[0], x.shape[1], x.shape[2]
    bufs = [(ctx.upload(w), ctx.upload(b)) for w, b in params]
    stages, ns = hip.conv_stages([(dw, db, c_out, post) for (dw, db), (_, c_out, post) in zip(bufs, spec)])
    kind = hip.hip.th_conv_chain_supported(c0, hw, hw, C.cast(stages, C.c_void_p), ns)
    assert kind == (want_kind if want_kind is not None else (1 if spec is REFERENCE else 2)), kind
    for _, _, post in spec:
        hw = hw // 2 if post == 1 else (1 if post == 2 else hw)
    c_last = spec[-1][1]
    y = ctx.empty(n * c_last * hw * hw)
    cnt = ctx.empty(n * c_last) if spec[-1][2] == 2 and want_cnt else None
    ctx.call("th_conv_chain_fwd", ctx.upload(x), C.cast(stages, C.c_void_p), ns, y, cnt, n, c0, x.shape[2], x.shape[2])
    ctx.sync()          # (the stage array and the uploads live until the launch has run)
    return y, cnt, hw, c_last


def _images(n, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, (n, 1, 28, 28)).astype(np.float32) / np.float32(255.0)


@pytest.mark.parametrize("n", [1, 3, 37, 256])
@pytest.mark.parametrize("name", ["reference", "simple"])
def test_chain_matches_the_oracle(ctx, O, name, n):
    spec = REFERENCE if name == "reference" else SIMPLE
    params = _params(spec, 11 + n)
    x = _images(n, n)
    ref, ref_cnt = _oracle_chain(O, x, spec, params)
    y, cnt, hw, c_last = _hip_chain(ctx, x, spec, params)
    got = ctx.download(y, (n, c_last, hw, hw))
    np.testing.assert_allclose(got, ref.reshape(got.shape), rtol=RTOL, atol=1e-5)
    if cnt is not None:
        got_cnt = ctx.download(cnt, (n, c_last))
        # a count can differ where a pre-activation sits within rounding of zero: allow a handful of planes to be off by a few elements
        off = np.abs(got_cnt - ref_cnt)
        assert off.max() <= 2 and (off > 0).mean() < 0.02, (off.max(), (off > 0).mean())


@pytest.mark.parametrize("name", ["reference", "simple"])
def test_chain_is_bit_identical_to_the_layered_launches_at_batch_256(ctx, name):
    spec = REFERENCE if name == "reference" else SIMPLE
    n = 256
    params = _params(spec, 5)
    x = _images(n, 99)
    ly, lcnt, hw = _hip_layered(ctx, x, spec, params)
    y, cnt, hw2, c_last = _hip_chain(ctx, x, spec, params)
    assert hw == hw2
    np.testing.assert_array_equal(ctx.download(y, (n, c_last, hw, hw)), ctx.download(ly, (n, c_last, hw, hw)))
    if cnt is not None:
        np.testing.assert_array_equal(ctx.download(cnt, (n, c_last)), ctx.download(lcnt, (n, c_last)))


def test_chain_without_counts_and_unsupported_stages(ctx):
    from taper_amd import hip
    params = _params(REFERENCE, 1)
    x = _images(2, 3)
    y, cnt, hw, c_last = _hip_chain(ctx, x, REFERENCE, params, want_cnt=False)
    y2, _, _, _ = _hip_chain(ctx, x, REFERENCE, params, want_cnt=True)
    np.testing.assert_array_equal(ctx.download(y, (2, c_last)), ctx.download(y2, (2, c_last)))
    # another channel count: no compiled instance -- the kernel that takes its stages as arguments (3)
    bufs = [(ctx.upload(w), ctx.upload(b)) for w, b in params]
    st, ns = hip.conv_stages([(bufs[0][0], bufs[0][1], 16, 1), (bufs[1][0], bufs[1][1], 64, 1)])
    assert hip.hip.th_conv_chain_supported(1, 28, 28, C.cast(st, C.c_void_p), ns) == 3
    st, ns = hip.conv_stages([(bufs[0][0], bufs[0][1], 32, 1), (bufs[1][0], bufs[1][1], 64, 1)])
    assert hip.hip.th_conv_chain_supported(1, 28, 28, C.cast(st, C.c_void_p), ns) == 2
    assert hip.hip.th_conv_chain_supported(1, 32, 32, C.cast(st, C.c_void_p), ns) == 0      # 32 planes of 32 x 32 + their pooled planes: past the LDS
    st16, ns16 = hip.conv_stages([(bufs[0][0], bufs[0][1], 16, 1), (bufs[1][0], bufs[1][1], 64, 1)])
    assert hip.hip.th_conv_chain_supported(1, 32, 32, C.cast(st16, C.c_void_p), ns16) == 3  # (r04: any square map up to 32 x 32 that fits)
    assert hip.hip.th_conv_chain_supported(3, 28, 28, C.cast(st, C.c_void_p), ns) == 0


@pytest.mark.parametrize("name", ["reference", "simple"])
def test_chain_propagates_non_finite_pixels_like_the_layered_launches(ctx, name):
    """Inf / NaN pixels: ReLU of NaN is 0 (`_mm_max_ps`, ops.rs:312-330), the pool's strict > never picks NaN (tensor.rs:1449-1461), and
    conv1's zero-weight tap padding on the matrix cores must not turn an Inf pixel into NaN (0 * Inf): the chain's outputs -- values AND
    where they are non-finite -- are those of the layer-by-layer launches"""
    spec = REFERENCE if name == "reference" else SIMPLE
    n = 256          # (the batch at which the layered launches take the same k order: bit-identical finite values)
    params = _params(spec, 21)
    x = _images(n, 5)
    rng = np.random.default_rng(8)
    for img in range(0, n, 3):
        r, c = rng.integers(0, 28, 2)
        x[img, 0, r, c] = [np.inf, -np.inf, np.nan][img % 3]
    x[1, 0, 27, 27] = np.inf          # the last pixel of an image: the single tile of waves 0 / 1
    ly, lcnt, hw = _hip_layered(ctx, x, spec, params)
    y, cnt, hw2, c_last = _hip_chain(ctx, x, spec, params)
    a, b = ctx.download(y, (n, c_last, hw, hw)), ctx.download(ly, (n, c_last, hw, hw))
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    # (ReLU turns NaN into 0 at every layer, so the planted values need not survive to the last stage: the comparison above is the point)


# ---- the kernel that takes its stages as arguments (th_conv_chain_supported == 3) ------------------------------------------------------------
@pytest.fixture
def generic(ctx):
    """the compiled instances switched off: their nets take the run-time-described kernel"""
    from taper_amd import hip
    hip.hip.th_debug_set_chain_generic(1)
    yield
    hip.hip.th_debug_set_chain_generic(0)


@pytest.mark.parametrize("name", ["reference", "simple"])
@pytest.mark.parametrize("n", [256, 5, 300])
def test_generic_chain_is_bit_identical_to_the_compiled_instances(ctx, name, n):
    """same per-output arithmetic (k order, bias after the sum, ReLU, strict-> maxima, plane sums): the run-time-described kernel gives the
    compiled instances' bits on their own nets; 300 images: workgroups walk more than one image"""
    from taper_amd import hip
    spec = REFERENCE if name == "reference" else SIMPLE
    params = _params(spec, 21)
    x = _images(n, 5 + n)
    y0, cnt0, hw, c_last = _hip_chain(ctx, x, spec, params)
    hip.hip.th_debug_set_chain_generic(1)
    try:
        y1, cnt1, hw1, _ = _hip_chain(ctx, x, spec, params, want_kind=3)
    finally:
        hip.hip.th_debug_set_chain_generic(0)
    assert hw == hw1
    np.testing.assert_array_equal(ctx.download(y1, (n, c_last, hw, hw)), ctx.download(y0, (n, c_last, hw, hw)))
    if cnt0 is not None:
        np.testing.assert_array_equal(ctx.download(cnt1, (n, c_last)), ctx.download(cnt0, (n, c_last)))


def _random_stage_list(rng):
    """a random run of Conv2dReLU / MaxPool2d(2) / global-average stages on a 28 x 28 single-channel image"""
    spec, c_in, hw = [], 1, 28
    for i in range(int(rng.integers(1, 6))):
        c_out = int(rng.choice([16, 32, 48, 64, 96, 128]))
        post = 1 if (hw % 2 == 0 and rng.random() < 0.5) else 0
        spec.append((c_in, c_out, post))
        c_in, hw = c_out, hw // 2 if post == 1 else hw
    last = spec[-1]
    end = 2 if (hw % 2 == 1 or rng.random() < 0.5) else 1
    spec[-1] = (last[0], last[1], end)
    return spec


@pytest.mark.parametrize("seed", range(24))
def test_generic_chain_random_stage_lists_match_the_oracle(ctx, O, seed):
    """random stage lists and batch sizes against the oracle's layer-by-layer tape ops (/root/reference/src/tensor.rs:1221-1285, 1391-1470,
    1524-1660); lists whose maps do not fit the LDS must be refused (th_conv_chain_supported == 0), everything else must run"""
    from taper_amd import hip
    rng = np.random.default_rng(900 + seed)
    spec = _random_stage_list(rng)
    n = int(rng.choice([1, 2, 7, 33, 96, 257]))
    params = _params(spec, 40 + seed)
    bufs = [(ctx.upload(w), ctx.upload(b)) for w, b in params]
    stages, ns = hip.conv_stages([(dw, db, c_out, post) for (dw, db), (_, c_out, post) in zip(bufs, spec)])
    kind = hip.hip.th_conv_chain_supported(1, 28, 28, C.cast(stages, C.c_void_p), ns)
    # what must fit: a stage's input planes beside its output (unless one round of tiles lets the output overlay the input)
    words, hw = 0, 28
    for c_in, c_out, post in spec:
        pad = lambda s: (s + 2) ** 2 + ((16 - ((s + 2) ** 2) % 32) + 32) % 32
        out = c_out * (pad(hw) if post == 0 else hw * hw + ((4 - (hw * hw) % 8) + 8) % 8)
        words = max(words, out, c_in * pad(hw))
        hw = hw // 2 if post == 1 else hw
    if kind == 0:
        assert words > 40960 // 2, (spec, words)        # only lists with a big map may be refused
        return
    assert kind in (2, 3)
    x = _images(n, seed)
    ref, ref_cnt = _oracle_chain(O, x, spec, params)
    y, cnt, hw, c_last = _hip_chain(ctx, x, spec, params, want_kind=kind)
    got = ctx.download(y, (n, c_last, hw, hw))
    np.testing.assert_allclose(got, ref.reshape(got.shape), rtol=RTOL, atol=1e-5 + RTOL * float(np.abs(ref).max()), err_msg=str(spec))
    if cnt is not None:
        off = np.abs(ctx.download(cnt, (n, c_last)) - ref_cnt)
        assert off.max() <= 2 and (off > 0).mean() < 0.02, (spec, off.max(), (off > 0).mean())


@pytest.mark.parametrize("c0,hw,spec", [(16, 14, [(16, 32, 0), (32, 64, 1)]), (32, 14, [(32, 32, 1), (32, 128, 2)]), (64, 7, [(64, 256, 2)]),
                                        (1, 14, [(1, 16, 1), (16, 16, 2)]), (1, 28, [(1, 16, 1), (16, 16, 1), (16, 512, 2)])])
def test_generic_chain_other_inputs(ctx, O, c0, hw, spec):
    """inputs that are not a 28 x 28 single-channel image: 16 / 32 / 64 input channels, 14 x 14 and 7 x 7 maps, 16 and 512 output channels
    (one channel tile; two rounds of tiles)"""
    rng = np.random.default_rng(c0 * 100 + hw)
    n = 9
    params = _params(spec, c0 + hw)
    x = rng.uniform(0, 1, (n, c0, hw, hw)).astype(np.float32)
    ref, ref_cnt = _oracle_chain(O, x, spec, params)
    y, cnt, hw_o, c_last = _hip_chain_general(ctx, x, spec, params)
    got = ctx.download(y, (n, c_last, hw_o, hw_o))
    np.testing.assert_allclose(got, ref.reshape(got.shape), rtol=RTOL, atol=1e-5 + RTOL * float(np.abs(ref).max()))


def _hip_chain_general(ctx, x, spec, params):
    from taper_amd import hip
    n, c0, s0 = x.shape[0], x.shape[1], x.shape[2]
    bufs = [(ctx.upload(w), ctx.upload(b)) for w, b in params]
    stages, ns = hip.conv_stages([(dw, db, c_out, post) for (dw, db), (_, c_out, post) in zip(bufs, spec)])
    assert hip.hip.th_conv_chain_supported(c0, s0, s0, C.cast(stages, C.c_void_p), ns) == 3
    hw = s0
    for _, _, post in spec:
        hw = hw // 2 if post == 1 else (1 if post == 2 else hw)
    c_last = spec[-1][1]
    y = ctx.empty(n * c_last * hw * hw)
    cnt = ctx.empty(n * c_last) if spec[-1][2] == 2 else None
    ctx.call("th_conv_chain_fwd", ctx.upload(x), C.cast(stages, C.c_void_p), ns, y, cnt, n, c0, s0, s0)
    ctx.sync()
    return y, cnt, hw, c_last


def test_generic_chain_refusals(ctx):
    from taper_amd import hip
    params = _params([(1, 32, 1), (32, 64, 1)], 1)
    bufs = [(ctx.upload(w), ctx.upload(b)) for w, b in params]
    f = hip.hip.th_conv_chain_supported
    st = lambda *rows: tuple(x if i == 0 else x for i, x in enumerate(hip.conv_stages([(bufs[0][0], bufs[0][1], c, p) for c, p in rows])))
    for rows, c0, hw, want in [([(24, 1), (64, 1)], 1, 28, 0),        # 24 output channels: not whole 16-channel tiles
                               ([(32, 1), (64, 0)], 1, 28, 3),        # ends in a conv row: its map is written (r04)
                               ([(32, 2), (64, 1)], 1, 28, 0),        # the global average is not the last stage
                               ([(32, 1), (64, 1), (64, 1)], 1, 28, 0),   # 7 x 7 cannot be pooled 2 x 2
                               ([(64, 0), (64, 1)], 1, 28, 0),        # 64 planes of 30 x 30 do not fit beside anything
                               ([(32, 1), (64, 1)], 3, 28, 0),        # 3 input channels
                               ([(16, 1), (32, 1)], 1, 32, 3),        # 32 x 32: the instance that takes the map size as an argument (r04)
                               ([(16, 1), (32, 1)], 1, 34, 0),        # 34 x 34: past the sizes the kernel takes
                               ([(16, 1), (16, 1)], 1, 30, 0),        # 30 -> 15 -> cannot be pooled 2 x 2
                               ([(16, 1), (32, 1)], 1, 28, 3)]:
        arr, ns = st(*rows)
        assert f(c0, hw, hw, C.cast(arr, C.c_void_p), ns) == want, rows


# ---- maps that are not MNIST-shaped (r04): any square input of 4 .. 32 pixels, runs that end in a conv row ----------------------------------
def _random_stage_list_any_size(rng):
    hw0 = int(rng.integers(8, 33))
    c0 = int(rng.choice([1, 1, 16, 32]))
    spec, c_in, hw = [], c0, hw0
    for i in range(int(rng.integers(1, 5))):
        c_out = int(rng.choice([16, 32, 48, 64]))
        post = 1 if (hw % 2 == 0 and hw >= 8 and rng.random() < 0.6) else 0
        spec.append((c_in, c_out, post))
        c_in, hw = c_out, hw // 2 if post == 1 else hw
    last = spec[-1]
    r = rng.random()
    end = 2 if r < 0.35 else (last[2] if r < 0.7 else 0)      # the global mean, whatever was drawn (a pool or a conv row), or a conv row
    spec[-1] = (last[0], last[1], end)
    return c0, hw0, spec


@pytest.mark.parametrize("seed", range(40))
def test_generic_chain_any_map_size_matches_the_oracle(ctx, O, seed):
    """random square inputs of 8 .. 32 pixels a side (1 / 16 / 32 channels), random stage lists incl. runs that end in a conv row, random batches,
    against the oracle's conv2d_relu / max_pool2d / avg_pool2d (/root/reference/src/tensor.rs:1221-1285, 1391-1470, 1524-1660); a list is only
    refused when a map does not fit the LDS"""
    from taper_amd import hip
    rng = np.random.default_rng(7000 + seed)
    c0, hw0, spec = _random_stage_list_any_size(rng)
    n = int(rng.choice([1, 3, 20, 130, 257]))
    params = _params(spec, 70 + seed)
    bufs = [(ctx.upload(w), ctx.upload(b)) for w, b in params]
    stages, ns = hip.conv_stages([(dw, db, c_out, post) for (dw, db), (_, c_out, post) in zip(bufs, spec)])
    kind = hip.hip.th_conv_chain_supported(c0, hw0, hw0, C.cast(stages, C.c_void_p), ns)
    pad = lambda s: (s + 2) ** 2 + ((16 - ((s + 2) ** 2) % 32) + 32) % 32
    words, hw = c0 * pad(hw0), hw0
    for c_in, c_out, post in spec:
        words = max(words, c_in * pad(hw) + c_out * (pad(hw) if post == 0 else hw * hw + 8))
        hw = hw // 2 if post == 1 else hw
    if kind == 0:
        assert words > 40960 // 2, (c0, hw0, spec, words)     # only lists with a big map may be refused
        return
    assert kind == 3, (kind, c0, hw0, spec)
    x = rng.uniform(0, 1, (n, c0, hw0, hw0)).astype(np.float32)
    ref, ref_cnt = _oracle_chain(O, x, spec, params)
    hw = hw0
    for _, _, post in spec:
        hw = hw // 2 if post == 1 else (1 if post == 2 else hw)
    c_last = spec[-1][1]
    y = ctx.empty(n * c_last * hw * hw)
    cnt = ctx.empty(n * c_last) if spec[-1][2] == 2 else None
    ctx.call("th_conv_chain_fwd", ctx.upload(x), C.cast(stages, C.c_void_p), ns, y, cnt, n, c0, hw0, hw0)
    ctx.sync()
    got = ctx.download(y, (n, c_last, hw, hw))
    np.testing.assert_allclose(got, ref.reshape(got.shape), rtol=RTOL, atol=1e-5 + RTOL * float(np.abs(ref).max()), err_msg=str((c0, hw0, spec)))
    if cnt is not None:
        off = np.abs(ctx.download(cnt, (n, c_last)) - ref_cnt)
        assert off.max() <= 2 and (off > 0).mean() < 0.02, (spec, off.max(), (off > 0).mean())


@pytest.mark.parametrize("hw0,spec", [(28, [(1, 32, 1), (32, 64, 0)]), (14, [(16, 32, 0)]), (7, [(64, 64, 0), (64, 128, 0)])])
def test_compiled_size_instances_write_a_conv_row_map(ctx, O, hw0, spec):
    """the 28 / 14 / 7 instances (map size compiled in) with a run that ends in a conv row"""
    from taper_amd import hip
    rng = np.random.default_rng(hw0)
    n, c0 = 11, spec[0][0]
    params = _params(spec, hw0)
    x = rng.uniform(0, 1, (n, c0, hw0, hw0)).astype(np.float32)
    ref, _ = _oracle_chain(O, x, spec, params)
    y, cnt, hw, c_last = _hip_chain_general(ctx, x, spec, params)
    got = ctx.download(y, (n, c_last, hw, hw))
    np.testing.assert_allclose(got, ref.reshape(got.shape), rtol=RTOL, atol=1e-5 + RTOL * float(np.abs(ref).max()))


@pytest.mark.parametrize("name", ["reference", "simple"])
@pytest.mark.parametrize("n", [257, 600, 1024])
def test_walking_chain_is_bit_identical_per_image(ctx, name, n):
    """more images than CUs (r05): min(n, 256) workgroups WALK the images -- the next image's pixels cross the fabric under the last k loop,
    the next first weight pass rides in the last pass -- with the per-image arithmetic untouched: image i of a batch of n > 256 gives the
    bits it gives as image (i mod 200) of batches of <= 200 (one workgroup per image, the kernel the oracle comparisons above hold)"""
    spec = REFERENCE if name == "reference" else SIMPLE
    from taper_amd._lib import hip as lib
    params = _params(spec, 21)
    x = _images(n, 1000 + n)
    lib.th_debug_set_chain_loop(1)              # (off by default: measured 3 - 5 % slower than one workgroup per image)
    try:
        y, cnt, hw, c_last = _hip_chain(ctx, x, spec, params)
    finally:
        lib.th_debug_set_chain_loop(-1)
    got = ctx.download(y, (n, c_last, hw, hw))
    got_cnt = ctx.download(cnt, (n, c_last)) if cnt is not None else None
    for lo in range(0, n, 200):
        hi = min(n, lo + 200)
        y1, cnt1, _, _ = _hip_chain(ctx, x[lo:hi], spec, params)
        np.testing.assert_array_equal(got[lo:hi], ctx.download(y1, (hi - lo, c_last, hw, hw)), err_msg=f"images {lo}..{hi}")
        if cnt is not None:
            np.testing.assert_array_equal(got_cnt[lo:hi], ctx.download(cnt1, (hi - lo, c_last)))


@pytest.mark.parametrize("n", [257, 320, 512, 600, 1024])
def test_two_workgroups_per_cu_instance_is_bit_identical_per_image(ctx, n):
    """from 257 images (more than one per CU) the simple chain runs its 128-register instance, two workgroups to a CU (conv_chain_simple_kernel<.., LEAN>: half-pass k
    loop, the same k order): image i of the batch gives the bits it gives in batches of <= 200 images (the one-per-CU instance, held to the
    oracle above)"""
    params = _params(SIMPLE, 33)
    x = _images(n, 2000 + n)
    y, cnt, hw, c_last = _hip_chain(ctx, x, SIMPLE, params)
    got = ctx.download(y, (n, c_last, hw, hw))
    for lo in range(0, n, 200):
        hi = min(n, lo + 200)
        y1, _, _, _ = _hip_chain(ctx, x[lo:hi], SIMPLE, params)
        np.testing.assert_array_equal(got[lo:hi], ctx.download(y1, (hi - lo, c_last, hw, hw)), err_msg=f"images {lo}..{hi}")
