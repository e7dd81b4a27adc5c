"""BASELINE configs[2] AT FULL SIZE (batch 256): the kernel instances `bench.py` and rocprofv3 show at that size are
chosen by workgroup count (`conv3x3_mfma_launch`: 4 channel tiles per workgroup once a launch has >= 384 of them, the
LDS-DMA staging, the fused 2x2 pool), so the small-batch parity cases never reach them.  Here every conv layer of both
CNNs runs at batch 256 against the oracle (`ot_conv2d`: /root/reference/src/tensor.rs:1221-1285,1728-1780, max-pool
1391-1470), with the launch configuration asserted through th_debug_last_conv_config, and both whole models run their
forward / backward / Adam steps at batch 256 -- eagerly AND through the Trainer's captured (fused) step, the form the
bench times -- against the oracle's tape (examples/train_mnist_cnn.rs:27,154-182)."""
import ctypes as C

import numpy as np
import pytest

from tests import backends

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


def last_conv_config(ctx):
    out = (C.c_int * 6)()
    ctx.call("th_debug_last_conv_config", C.cast(out, C.c_void_p))
    return dict(ct=out[0], dma=out[1], waves=4 * out[2], grid=(out[3], out[4]), pool=out[5])


# n, c_in, hw, c_out, then what the launch must pick at this size: the image-resident kernel's (images per unit, channel tiles,
# pixel tiles per wave) -- the instances rocprofv3 lists for the batch-256 CNN steps -- and the 128-pixel kernel's channel tiles
FULL_LAYERS = [
    (256, 32, 28, 32, (1, 2, 13), 2),    # reference CNN conv2 (+ pool): one 28x28 image x 32 channels per CU
    (256, 32, 14, 64, (4, 1, 7), 4),     # reference CNN conv3; simple CNN conv2 (+ pool): four 14x14 images x 16 channels
    (256, 64, 14, 64, (4, 1, 7), 4),     # reference CNN conv4 (+ pool)
    (256, 64, 7, 128, (8, 1, 4), 2),     # reference CNN conv5: eight 7x7 images x 16 channels
]


@pytest.mark.parametrize("kernel", ["layer_chain", "image_resident", "pixel_block"])
@pytest.mark.parametrize("n,c_in,hw,c_out,img_cfg,ct", FULL_LAYERS)
def test_conv3x3_full_size_layers(ctx, O, n, c_in, hw, c_out, img_cfg, ct, kernel):
    """the default choice at batch 256 (r04): the layer as a one-stage chain with the chain's compiled tile mapping (conv_layer_chain_kernel,
    config id 8); th_debug_set_conv_img forces the image-resident kernel (1) and the 128-pixel kernel (0) onto the same shapes"""
    ctx.call("th_debug_set_conv_img", {"layer_chain": -1, "image_resident": 1, "pixel_block": 0}[kernel])
    try:
        _full_size_layer(ctx, O, n, c_in, hw, c_out, {"layer_chain": ("chain", c_out // 16), "image_resident": img_cfg, "pixel_block": None}[kernel], ct)
    finally:
        ctx.call("th_debug_set_conv_img", -1)


def _expect_cfg(cfg, img_cfg, ct, pool):
    if img_cfg is not None and img_cfg[0] == "chain":
        assert cfg["dma"] == 8 and cfg["ct"] == img_cfg[1] and cfg["grid"][0] == 256 and cfg["pool"] == pool, cfg
    elif img_cfg is None:
        assert cfg["ct"] == ct and cfg["dma"] == 1 and cfg["pool"] == pool, cfg
        assert cfg["grid"][0] * cfg["grid"][1] >= 256, cfg                 # a chip-filling launch, not the small-batch shape
    else:
        img, ict, tpw = img_cfg
        assert cfg["dma"] in (2, 3, 4, 5) and cfg["ct"] == ict and cfg["waves"] == 4 * tpw and cfg["grid"][1] == img and cfg["pool"] == pool, cfg   # 2 / 3: 8 / 4 waves
        assert cfg["grid"][0] == 256, cfg                                  # one unit per CU


def _full_size_layer(ctx, O, n, c_in, hw, c_out, img_cfg, ct):
    rng = np.random.default_rng(c_in * 1000 + hw * 10 + c_out)
    x = rng.uniform(-1, 1, (n, c_in, hw, hw)).astype(np.float32)
    bound = np.sqrt(6.0 / (c_in * 9))                                   # nn.rs:219-222
    wt = rng.uniform(-bound, bound, (c_out, c_in, 3, 3)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, c_out).astype(np.float32)
    xt, wtt, bt = O.Tensor(x), O.Tensor(wt), O.Tensor(b)
    ref = xt.conv2d_relu(wtt, bt, (1, 1), (1, 1), (1, 1))
    ref_d = ref.data()
    dx, dw, db = ctx.upload(x), ctx.upload(wt), ctx.upload(b)
    y = ctx.empty(n * c_out * hw * hw)
    ctx.call("th_conv3x3_fwd", dx, dw, db, y, n, c_in, hw, hw, c_out, 1, 0, 1)
    _expect_cfg(last_conv_config(ctx), img_cfg, ct, 0)
    got = ctx.download(y, ref_d.shape)
    np.testing.assert_allclose(got, ref_d, rtol=RTOL, atol=1e-5)
    # no ReLU, no bias (the pre-activation itself)
    ctx.call("th_conv3x3_fwd", dx, dw, None, y, n, c_in, hw, hw, c_out, 1, 0, 0)
    ref2 = xt.conv2d(wtt, None, (1, 1), (1, 1), (1, 1)).data()
    np.testing.assert_allclose(ctx.download(y, ref2.shape), ref2, rtol=RTOL, atol=1e-5)
    # fused 2x2 / stride-2 max-pool epilogue vs the oracle's conv2d_relu -> max_pool2d (values; the fused form keeps no indices)
    if hw % 2 == 0:
        assert ctx_supported(ctx, c_in, hw, c_out)
        yp = ctx.empty(n * c_out * (hw // 2) ** 2)
        ctx.call("th_conv3x3_pool2_fwd", dx, dw, db, yp, n, c_in, hw, hw, c_out, 1, 1)
        _expect_cfg(last_conv_config(ctx), img_cfg, ct, 1)
        pooled = ref.max_pool2d((2, 2), (2, 2), (0, 0)).data()
        np.testing.assert_allclose(ctx.download(yp, pooled.shape), pooled, rtol=RTOL, atol=1e-5)
        # and bit-identical to the unfused HIP pair (same fmaf chain, max is exact)
        ctx.call("th_conv3x3_fwd", dx, dw, db, y, n, c_in, hw, hw, c_out, 1, 0, 1)
        np.testing.assert_array_equal(ctx.download(yp, pooled.shape), got.reshape(n, c_out, hw // 2, 2, hw // 2, 2).max(axis=(3, 5)))


def ctx_supported(ctx, c_in, hw, c_out):
    from taper_amd import hip
    return hip.hip.th_conv3x3_pool2_supported(c_in, hw, hw, c_out, 1) == 1


def test_conv1_full_size(ctx, O):
    """conv1 of both CNNs at batch 256 (single input channel: the dedicated kernels, plain and pooled)"""
    rng = np.random.default_rng(1)
    n = 256
    x = (rng.integers(0, 256, (n, 1, 28, 28)).astype(np.float32) / np.float32(255.0))
    wt = rng.uniform(-0.8, 0.8, (32, 1, 3, 3)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, 32).astype(np.float32)
    ref = O.Tensor(x).conv2d_relu(O.Tensor(wt), O.Tensor(b), (1, 1), (1, 1), (1, 1))
    dx, dw, db = ctx.upload(x), ctx.upload(wt), ctx.upload(b)
    y = ctx.empty(n * 32 * 28 * 28)
    ctx.call("th_conv3x3_fwd", dx, dw, db, y, n, 1, 28, 28, 32, 1, 0, 1)
    np.testing.assert_allclose(ctx.download(y, ref.shape()), ref.data(), rtol=RTOL, atol=1e-5)
    yp = ctx.empty(n * 32 * 14 * 14)
    ctx.call("th_conv3x3_pool2_fwd", dx, dw, db, yp, n, 1, 28, 28, 32, 1, 1)
    pooled = ref.max_pool2d((2, 2), (2, 2), (0, 0)).data()
    np.testing.assert_allclose(ctx.download(yp, pooled.shape), pooled, rtol=RTOL, atol=1e-5)


def test_maxpool_full_size_bit_exact(ctx, O):
    """[256,32,28,28] -> [256,32,14,14] (tensor.rs:1391-1470): values and absolute flat indices bit-exact, then the backward"""
    rng = np.random.default_rng(2)
    n, c, hw = 256, 32, 28
    x = rng.integers(-6, 7, (n, c, hw, hw)).astype(np.float32)      # many ties: the first maximum must win
    y_ref, idx_ref = O.Tensor(x).max_pool2d((2, 2), (2, 2), (0, 0), return_indices=True)
    dx = ctx.upload(x)
    y, idx = ctx.empty(n * c * 14 * 14), ctx.empty(n * c * 14 * 14, np.int64)
    ctx.call("th_maxpool2d_fwd", dx, y, idx, n, c, hw, hw, 2, 2, 2, 2, 0, 0)
    np.testing.assert_array_equal(ctx.download(y, y_ref.shape()), y_ref.data())
    np.testing.assert_array_equal(ctx.download(idx, idx_ref.shape, np.int64), idx_ref)


MODELS = {"cnn_simple": backends.cnn_simple, "cnn_reference": backends.cnn_reference}
# observed on MI355X (profiles/r03_parity_margins.json) x 2: see _training_steps_parity
LOSS_RTOL = 2e-6            # observed 7.4e-7 of the largest loss
STEP_ERR_OVER_LR = 1.5e-2   # observed 7.4e-3 (simple CNN, the classifier's weight), 3.0e-3 (reference CNN)


def _grads_close(h_grads, o_grads):
    assert len(h_grads) == len(o_grads)
    for i, (hg, og) in enumerate(zip(h_grads, o_grads)):
        assert (hg is None) == (og is None), f"param {i}: grad None-ness differs"
        if og is not None:
            scale = float(np.abs(og).max())
            np.testing.assert_allclose(hg, og, rtol=RTOL, atol=2e-4 * scale + 1e-8, err_msg=f"param {i}")


@pytest.mark.parametrize("name", ["cnn_simple", "cnn_reference"])
@pytest.mark.parametrize("fuse", [True, False])
def test_cnn_forward_backward_parity_batch_256(name, fuse):
    """examples/train_mnist_cnn.rs:27 -- the whole model at its real batch: logits, loss, accuracy, every gradient
    (and which ones are None, Q2) against the oracle's tape"""
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(256 + len(name))
    batch = 256
    spec = backends.nonzero_biases(MODELS[name](rng), rng)
    x, y = backends.mnist_like(rng, batch)
    shape = (batch, 1, 28, 28)
    hm, om = H.sequential(spec, fuse=fuse), Orc.sequential(spec)
    h_loss, h_acc, h_logits, h_grads = H.forward_backward(hm, x, y, shape)
    o_loss, o_acc, o_logits, o_grads = Orc.forward_backward(om, x, y, shape)
    np.testing.assert_allclose(h_logits, o_logits, rtol=RTOL, atol=RTOL * float(np.abs(o_logits).max()))
    assert abs(h_loss - o_loss) <= RTOL * max(1.0, abs(o_loss))
    assert abs(h_acc - o_acc) <= 1.0 / batch
    _grads_close(h_grads, o_grads)
    assert h_grads[0] is None and o_grads[0] is None     # Q2: conv weights never receive gradients


@pytest.mark.parametrize("name", ["cnn_simple", "cnn_reference"])
@pytest.mark.parametrize("mode", ["eager", "graph", "graph_layered"])
def test_cnn_training_steps_parity_batch_256(name, mode):
    """3 Adam steps (lr 1e-2, wd 1e-4: train_mnist_cnn.rs:108-109) at batch 256 -- `graph` is the Trainer's captured,
    fused step (conv + pool in one launch, pooled bias gradients, fused classifier tail, Adam in the epilogues): exactly
    what bench.py times for the CNN workloads -- per-step loss / hit count and every weight against the oracle.  `graph` launches the
    convolutional front as ONE kernel (th_conv_chain_fwd), `graph_layered` layer by layer (T.set_conv_chain(False))"""
    import taper_amd as T
    T.set_conv_chain(mode != "graph_layered")
    try:
        _training_steps_parity(T, name, mode)
    finally:
        T.set_conv_chain(True)


def _training_steps_parity(T, name, mode):
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(11 + len(name))
    batch, steps, lr = 256, 3, 1e-2
    spec = backends.nonzero_biases(MODELS[name](rng), rng)
    hm, om = H.sequential(spec), Orc.sequential(spec)
    hopt = T.Adam(hm.parameters(), lr, None, None, 1e-4)
    oopt = Orc.m.Adam(om.parameters(), lr, None, None, 1e-4)
    tr = T.Trainer(hm, hopt, sample_shape=(1, 28, 28))
    x, y = backends.mnist_like(rng, steps * batch)
    ref = [om.train_step(oopt, x[s * batch:(s + 1) * batch], y[s * batch:(s + 1) * batch], (batch, 1, 28, 28)) for s in range(steps)]
    calls0 = mlp3_calls()
    if mode != "eager":
        ep = tr.run_epoch(T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False), T.Trainer.GRAPH)
        losses, ncorrect = ep["losses"], ep["ncorrect"]
        cfg = last_conv_config_host()
        if mode == "graph":
            # the conv chain ran in this process's step (7: with the simple CNN's classifier rows in its last epilogue, th_conv_chain_head_fwd)
            # (9: with the reference CNN's three-layer classifier's rows in the chain launch, th_conv_chain_mlp3_xent: r05)
            assert cfg["dma"] == (9 if name == "cnn_reference" else 7) and cfg["ct"] == (1 if name == "cnn_reference" else 2), cfg
        else:
            assert cfg["dma"] in (2, 3, 4, 5, 8), cfg   # a layer-by-layer matrix-core conv ran in this process's step (8: as a one-stage chain)
            if name == "cnn_reference":     # its three-layer classifier took th_mlp3_xent (two launches) in the captured step
                assert mlp3_calls() > calls0
    else:
        losses, ncorrect = [], []
        for s in range(steps):
            l, a = tr.train_step(T.Tensor(x[s * batch:(s + 1) * batch]), T.Tensor(y[s * batch:(s + 1) * batch]))
            losses.append(l)
            ncorrect.append(a * batch)
    from tests import margins
    tag = f"{name}_b256_3_adam_steps[{mode}]"
    margins.record(tag, "losses", losses, [r["loss"] for r in ref])
    np.testing.assert_allclose(losses, [r["loss"] for r in ref], rtol=0, atol=LOSS_RTOL * max(abs(r["loss"]) for r in ref))
    assert np.abs(np.asarray(ncorrect) - np.asarray([r["acc"] * batch for r in ref])).max() <= 1.5
    for i, (hp, op) in enumerate(zip(hm.parameters(), om.parameters())):
        m = margins.record(tag, f"param{i}", hp.data(), op.data(), lr=lr)
        # Adam's first steps move a weight by lr * m / (sqrt(v) + eps) ~ lr * sign(g): the natural scale of a weight error after 3 steps is
        # lr, not the weight ("parity margins"); the bound is 2x the largest error observed on MI355X (profiles/r03_parity_margins.json)
        assert m["err_over_lr"] <= STEP_ERR_OVER_LR, (i, m)
    assert hopt.t() == steps


FULL_BWD_STEP0_LOSS = 4e-6     # the first step's loss (identical weights), of its size: the faithful steps' LOSS_RTOL class
FULL_BWD_STEP_LOSS = 1.6e-4    # later steps' losses, of the largest: observed 7.8e-5 (reference CNN, step 2) -- EVERY conv weight has moved by ~lr sign(g)
                               # by then, and the elements whose gradient is a cancellation down to ~eps move differently under another summation order
FULL_BWD_STEP_ERR_OVER_LR = 4e-2   # weights after the steps, in units of lr, on all but FULL_BWD_OUTLIERS of a tensor's elements
FULL_BWD_OUTLIERS = 2e-2           # (elements whose gradient is small against one flipped pixel's contribution)
FULL_BWD_M = 2.5e-2                 # Adam's first moment after the steps, of the tensor's scale: observed 1.23e-2 (reference CNN, fc1 after 2 steps at lr 1e-2: step 2 runs on conv weights that each moved by ~lr sign(g)); 2x


@pytest.mark.parametrize("name,steps", [("cnn_simple", 3), ("cnn_reference", 2)])
def test_cnn_full_backward_graph_steps_parity_batch_256(name, steps):
    """`full_backward` (extension: every conv weight trains; the reference cuts the tape, Q2) through the TRAINER at batch 256, captured --
    the configuration bench.py quotes `full_bwd_ms` on (step 0 runs eagerly, the others are replayed graphs): per-step loss / hit count and
    every weight after the Adam steps against the oracle's differentiable im2col chain (oracle conv_mode 1) -- the kernel-level
    comparisons stop at batch 136, eager (tests/test_gpu_step.py)"""
    import taper_amd as T
    from tests import margins
    H, Orc = backends.get("hip"), backends.get("oracle")
    Orc.set_zero_sentinel(True)
    rng = np.random.default_rng(23 + len(name))
    batch, lr = 256, 1e-2
    spec = backends.nonzero_biases(MODELS[name](rng), rng)
    x, y = backends.mnist_like(rng, steps * batch)
    om = Orc.sequential(spec, full_backward=True)
    oopt = Orc.m.Adam(om.parameters(), lr, None, None, 1e-4)
    ref = [om.train_step(oopt, x[s * batch:(s + 1) * batch], y[s * batch:(s + 1) * batch], (batch, 1, 28, 28)) for s in range(steps)]
    try:
        hm = H.sequential(spec, full_backward=True)
        hopt = T.Adam(hm.parameters(), lr, None, None, 1e-4)
        tr = T.Trainer(hm, hopt, sample_shape=(1, 28, 28))
        ep = tr.run_epoch(T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False), T.Trainer.GRAPH)
    finally:
        T.set_full_backward(False)
    tag = "test_cnn_full_backward_graph_steps_parity_batch_256"
    margins.check(f"{name}_loss_step0", ep["losses"][:1], [ref[0]["loss"]], FULL_BWD_STEP0_LOSS, test=tag)
    margins.check(f"{name}_losses", ep["losses"], [r["loss"] for r in ref], FULL_BWD_STEP_LOSS, test=tag)
    assert np.abs(np.asarray(ep["ncorrect"]) - np.asarray([r["acc"] * batch for r in ref])).max() <= 1.5
    assert hopt.t() == steps
    # Every conv weight trains here, behind ReLU masks and pool arg-maxima over 6.4 M activations per layer: a pre-activation within rounding of
    # zero (or two window elements within rounding of each other) flips under another summation order and moves the gradients it feeds by one
    # pixel's contribution -- discontinuities of the function, several per step at this size -- and Adam turns a small gradient's sign into a
    # step of ~lr.  So: Adam's first moment (linear in the gradients of both steps) against the oracle's within FULL_BWD_M of its scale; the
    # weights within 2 lr per step everywhere (a flipped sign) and within FULL_BWD_STEP_ERR_OVER_LR of lr on all but FULL_BWD_OUTLIERS of
    # the elements -- all recorded (tests/margins.py).
    hm_m, _ = hopt.moments()
    off = 0
    for i, (hp, op) in enumerate(zip(hm.parameters(), om.parameters())):
        got, ref = np.ravel(hp.data()).astype(np.float64), np.ravel(op.data()).astype(np.float64)
        assert float(np.abs(got - np.ravel(spec_param(spec, i))).max()) > 0, f"param {i} did not train in full_backward mode"
        margins.check(f"{name}_m{i}", hm_m[off:off + got.size], oopt.m(i), FULL_BWD_M, test=tag)
        off += got.size
        err = np.abs(got - ref) / lr
        margins.check(f"{name}_param{i}", got, ref, 2.0 * steps, lr=lr, test=tag)
        outliers = float((err > FULL_BWD_STEP_ERR_OVER_LR).mean())
        margins.record(tag, f"{name}_param{i}_outlier_fraction", [outliers], [0.0])
        assert outliers <= FULL_BWD_OUTLIERS, f"param {i}: {outliers:.4f} of the elements are more than {FULL_BWD_STEP_ERR_OVER_LR} lr away"


def spec_param(spec, i):
    """the i-th parameter's initial value in a backends spec (w, b per layer that has them)"""
    flat = [l[k] for l in spec if "w" in l for k in ("w", "b") if l.get(k) is not None]
    return flat[i]


@pytest.mark.parametrize("name", ["cnn_simple", "cnn_reference"])
def test_conv_chain_step_is_bit_identical_to_the_layered_step(name):
    """the Trainer's captured step with the convolutional front as one launch against the same step launched layer by layer: same
    per-output arithmetic, so every weight and every loss agrees bit for bit after 3 steps at batch 256 (the classifier as its own
    launches in both: with its rows inside the chain launch the logits add in another order -- tests/test_gpu_chain_head.py)"""
    import taper_amd as T
    H = backends.get("hip")
    batch, steps = 256, 3
    out = []
    for chain in (True, False):
        rng = np.random.default_rng(77)
        spec = backends.nonzero_biases(MODELS[name](rng), rng)
        x, y = backends.mnist_like(rng, steps * batch)
        T.set_conv_chain(chain)
        T.set_conv_chain_head(False)
        try:
            hm = H.sequential(spec)
            opt = T.Adam(hm.parameters(), 1e-2, None, None, 1e-4)
            tr = T.Trainer(hm, opt, sample_shape=(1, 28, 28))
            ep = tr.run_epoch(T.DataLoader(T.MNISTDataset.from_host(x, y), batch, False), T.Trainer.GRAPH)
            assert (last_conv_config_host()["dma"] == 6) == chain
            out.append((ep["losses"], [p.data() for p in hm.parameters()]))
        finally:
            T.set_conv_chain(True)
            T.set_conv_chain_head(True)
    np.testing.assert_array_equal(out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_array_equal(a, b)


def mlp3_calls():
    from taper_amd import hip
    n = C.c_int64(0)
    hip.hip.th_debug_mlp3_calls(C.byref(n))
    return n.value


def last_conv_config_host():
    """the host library enqueues on its own context but on the calling thread: same thread-local record"""
    import taper_amd as T
    from taper_amd import hip
    c = hip.Ctx(handle=T.Device.ctx_handle())
    return last_conv_config(c)


SMALL_IMG_CASES = [  # n, c_in, h, w, c_out, pad: the image-resident kernel FORCED onto small / ragged launches (it normally needs >= 128 units)
    (3, 32, 28, 28, 32, 1), (5, 32, 14, 14, 64, 1), (9, 64, 7, 7, 128, 1), (2, 8, 10, 12, 16, 1), (3, 16, 9, 11, 33, 0), (7, 24, 5, 5, 150, 1),
    (1, 64, 1, 1, 10, 1), (17, 8, 3, 3, 4, 0), (2, 40, 20, 20, 48, 1), (6, 8, 6, 8, 20, 1), (1, 8, 28, 28, 7, 1), (33, 16, 7, 7, 16, 1),
]


@pytest.mark.parametrize("n,c_in,h,w,c_out,pad", SMALL_IMG_CASES)
@pytest.mark.parametrize("relu", [0, 1])
def test_image_resident_kernel_on_small_and_ragged_shapes(ctx, O, n, c_in, h, w, c_out, pad, relu):
    """ragged image groups (n not a multiple of the images per unit), channel counts that are not multiples of 16, pad 0, 1x1 planes,
    partial pixel tiles: against the oracle, and bit-identical to the 128-pixel kernel (same fmaf chains in the same k order)"""
    rng = np.random.default_rng(n * 131 + c_in * 17 + h * 5 + c_out)
    x = rng.uniform(-1, 1, (n, c_in, h, w)).astype(np.float32)
    wt = rng.uniform(-0.5, 0.5, (c_out, c_in, 3, 3)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, c_out).astype(np.float32)
    xt, wtt, bt = O.Tensor(x), O.Tensor(wt), O.Tensor(b)
    ref = (xt.conv2d_relu if relu else xt.conv2d)(wtt, bt, (1, 1), (pad, pad), (1, 1))
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    dx, dw, db = ctx.upload(x), ctx.upload(wt), ctx.upload(b)
    out = {}
    try:
        for mode in (1, 0):
            ctx.call("th_debug_set_conv_img", mode)
            y = ctx.empty(n * c_out * ho * wo)
            ctx.call("th_conv3x3_fwd", dx, dw, db, y, n, c_in, h, w, c_out, pad, 0, relu)
            cfg = last_conv_config(ctx)
            if mode == 1:
                assert cfg["dma"] in (2, 3, 4, 5), cfg
            else:
                assert cfg["dma"] not in (2, 3, 4, 5), cfg
            out[mode] = ctx.download(y, (n, c_out, ho, wo))
            pooled = None
            if ho % 2 == 0 and wo % 2 == 0 and c_out % 4 == 0 and ctx_supported_pad(c_in, h, w, c_out, pad):
                yp = ctx.empty(n * c_out * (ho // 2) * (wo // 2))
                ctx.call("th_conv3x3_pool2_fwd", dx, dw, db, yp, n, c_in, h, w, c_out, pad, relu)
                pooled = ctx.download(yp, (n, c_out, ho // 2, wo // 2))
                np.testing.assert_array_equal(pooled, out[mode].reshape(n, c_out, ho // 2, 2, wo // 2, 2).max(axis=(3, 5)))
    finally:
        ctx.call("th_debug_set_conv_img", -1)
    np.testing.assert_allclose(out[1], ref.data(), rtol=RTOL, atol=1e-5)
    np.testing.assert_array_equal(out[1], out[0])


def ctx_supported_pad(c_in, h, w, c_out, pad):
    from taper_amd import hip
    return hip.hip.th_conv3x3_pool2_supported(c_in, h, w, c_out, pad) == 1


@pytest.mark.parametrize("n,c_in,h,w,c_out,pad", [(256, 64, 7, 7, 128, 1), (9, 64, 7, 7, 128, 1), (5, 8, 6, 8, 20, 1), (33, 16, 5, 5, 16, 0),
                                                   (2, 32, 14, 14, 64, 1)])
def test_conv3x3_relu_global_avgpool_fused(ctx, O, n, c_in, h, w, c_out, pad):
    gap_case(ctx, O, n, c_in, h, w, c_out, pad)


def ctx_supported_gap(n, c_in, h, w, c_out, pad):
    from taper_amd import hip
    return hip.hip.th_conv3x3_gap_supported(n, c_in, h, w, c_out, pad) == 1


def gap_case(ctx, O, n, c_in, h, w, c_out, pad):
    """th_conv3x3_gap_fwd (Conv2dReLU -> global average pool in one launch: the reference CNN's conv5 + AdaptiveAvgPool2d((1, 1)),
    examples/train_mnist_cnn.rs:73-84): plane means and per-plane counts of positive outputs against the oracle's conv2d_relu -> avg_pool2d,
    and bit-identical to the unfused HIP pair"""
    from taper_amd import hip
    assert hip.hip.th_conv3x3_gap_supported(n, c_in, h, w, c_out, pad) == 1
    rng = np.random.default_rng(n + c_in + h * 3 + c_out)
    x = rng.uniform(-1, 1, (n, c_in, h, w)).astype(np.float32)
    bound = np.sqrt(6.0 / (c_in * 9))
    wt = rng.uniform(-bound, bound, (c_out, c_in, 3, 3)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, c_out).astype(np.float32)
    ho, wo = h + 2 * pad - 2, w + 2 * pad - 2
    ref_map = O.Tensor(x).conv2d_relu(O.Tensor(wt), O.Tensor(b), (1, 1), (pad, pad), (1, 1))
    ref_mean = ref_map.avg_pool2d((ho, wo), (ho, wo), (0, 0)).data().reshape(n, c_out)
    ref_cnt = (ref_map.data() > 0).sum(axis=(2, 3)).astype(np.float32)
    dx, dw, db = ctx.upload(x), ctx.upload(wt), ctx.upload(b)
    ym, cnt = ctx.empty(n * c_out), ctx.empty(n * c_out)
    ctx.call("th_conv3x3_gap_fwd", dx, dw, db, ym, cnt, n, c_in, h, w, c_out, pad, 1)
    cfg = last_conv_config(ctx)
    assert cfg["dma"] in (2, 3, 4, 5, 8) and cfg["pool"] == 1, cfg     # (8: the reference CNN's conv5 shape takes the one-stage chain kernel)
    got_mean, got_cnt = ctx.download(ym, (n, c_out)), ctx.download(cnt, (n, c_out))
    np.testing.assert_allclose(got_mean, ref_mean, rtol=RTOL, atol=1e-5)
    # a count differs from the oracle's only where an output sits within rounding of 0
    assert np.abs(got_cnt - ref_cnt).max() <= 1 and (got_cnt != ref_cnt).mean() < 0.01
    # the unfused HIP pair, same kernel family: bit-identical means and counts
    ctx.call("th_debug_set_conv_img", 1)
    try:
        y = ctx.empty(n * c_out * ho * wo)
        ctx.call("th_conv3x3_fwd", dx, dw, db, y, n, c_in, h, w, c_out, pad, 0, 1)
    finally:
        ctx.call("th_debug_set_conv_img", -1)
    ym2, cnt2 = ctx.empty(n * c_out), ctx.empty(n * c_out)
    ctx.call("th_avgpool2d_global_fwd_counts", y, ym2, cnt2, n, c_out, ho * wo)
    np.testing.assert_array_equal(got_cnt, ctx.download(cnt2, (n, c_out)))
    np.testing.assert_array_equal(got_mean, ctx.download(ym2, (n, c_out)))
    # without the counts
    ctx.call("th_conv3x3_gap_fwd", dx, dw, db, ym2, None, n, c_in, h, w, c_out, pad, 1)
    np.testing.assert_array_equal(got_mean, ctx.download(ym2, (n, c_out)))
