"""MNIST IDX ingestion (`/root/reference/src/data/mnist.rs:184-274`) -- the host library's
`MNISTDataset::from_idx` against the oracle's restatement, on the first 2048 labels of the reference's own
`data/mnist/train_labels` (tests/golden/mnist_train_labels_head.idx1; the reference checkout has no image blobs, so the
image file is synthesised here in the same IDX3 format).  Header / size errors are raised before any device call, so
those cases run without a GPU."""
import struct
from pathlib import Path

import numpy as np
import pytest

from oracle import train_extra as OX

GOLDEN = Path(__file__).resolve().parent / "golden"
LABELS = GOLDEN / "mnist_train_labels_head.idx1"
MNIST_FIRST_LABELS = [5, 0, 4, 1, 9, 2, 1, 3, 1, 4, 3, 5, 3, 6, 1, 7, 2, 8, 6, 9]   # the well-known head of MNIST's training labels


def _images_idx(px: np.ndarray, rows=28, cols=28, magic=0x803, count=None) -> bytes:
    return struct.pack(">IIII", magic, px.shape[0] if count is None else count, rows, cols) + px.astype(np.uint8).tobytes()


def test_oracle_idx_labels_on_the_reference_data_file():
    """pins the oracle's label parser on real data: header, count, u8 -> f32"""
    y = OX.load_idx_labels(LABELS.read_bytes())
    assert y.dtype == np.float32 and y.shape == (2048,)
    assert y[:20].astype(int).tolist() == MNIST_FIRST_LABELS
    assert set(np.unique(y).astype(int)) == set(range(10))
    counts = np.bincount(y.astype(int), minlength=10)
    assert counts.sum() == 2048 and counts.min() > 150      # all ten digits, roughly balanced, like MNIST


def test_oracle_idx_images_values_and_errors():
    rng = np.random.default_rng(0)
    px = rng.integers(0, 256, (7, 784), dtype=np.uint8)
    x = OX.load_idx_images(_images_idx(px))
    assert x.shape == (7, 784) and x.dtype == np.float32
    assert np.array_equal(x, px.astype(np.float32) / np.float32(255.0))          # mnist.rs:226, f32 division
    assert x.min() >= 0.0 and x.max() <= 1.0
    for bad, text in [(_images_idx(px, magic=0x801), "Invalid magic number for images: 0x801"),
                      (_images_idx(px, rows=27), "Unexpected image size: 27x28"),
                      (_images_idx(px, count=8), "File size mismatch. Expected 6288, got 5504"),
                      (b"\0" * 15, "too small")]:
        with pytest.raises(ValueError, match=text):
            OX.load_idx_images(bad)
    for bad, text in [(struct.pack(">II", 0x803, 3) + b"\1\2\3", "Invalid magic number for labels: 0x803"),
                      (struct.pack(">II", 0x801, 4) + b"\1\2\3", "File size mismatch. Expected 12, got 11"),
                      (b"\0" * 7, "too small")]:
        with pytest.raises(ValueError, match=text):
            OX.load_idx_labels(bad)


ERROR_CASES = [
    # (images bytes builder, labels bytes builder, text the host error must carry)  -- mnist.rs:190-217, 242-262
    ("img_magic", "Invalid magic number for images"),
    ("img_small", "too small"),
    ("img_rows", "Unexpected image size"),
    ("img_size", "File size mismatch"),
    ("lab_magic", "Invalid magic number for labels"),
    ("lab_small", "too small"),
    ("lab_size", "File size mismatch"),
    ("count_mismatch", "count mismatch"),
    ("missing", "Failed to open"),
]


@pytest.mark.parametrize("case,text", ERROR_CASES, ids=[c for c, _ in ERROR_CASES])
def test_from_idx_error_paths(tmp_path, case, text):
    """every `Err(...)` of load_images / load_labels surfaces as taper_amd.Error with the reference's message stem
    (no device call happens before the files validate, so this runs on the CPU-only box too)"""
    import taper_amd as T
    rng = np.random.default_rng(1)
    n = 16
    px = rng.integers(0, 256, (n, 784), dtype=np.uint8)
    img = _images_idx(px)
    lab = struct.pack(">II", 0x801, n) + bytes(rng.integers(0, 10, n, dtype=np.uint8))
    if case == "img_magic":
        img = _images_idx(px, magic=0x802)
    elif case == "img_small":
        img = img[:12]
    elif case == "img_rows":
        img = _images_idx(px, rows=14)
    elif case == "img_size":
        img = img[:-1]
    elif case == "lab_magic":
        lab = struct.pack(">II", 0x803, n) + lab[8:]
    elif case == "lab_small":
        lab = lab[:6]
    elif case == "lab_size":
        lab = lab + b"\0"
    elif case == "count_mismatch":
        lab = struct.pack(">II", 0x801, n - 1) + lab[8:-1]
    ip, lp = tmp_path / "images", tmp_path / "labels"
    if case != "missing":
        ip.write_bytes(img)
    lp.write_bytes(lab)
    with pytest.raises(T.TaperError, match=text):
        T.MNISTDataset.from_idx(ip, lp)


@pytest.mark.gpu
def test_from_idx_round_trip_matches_oracle(tmp_path):
    """IDX files -> device-resident dataset -> read back: images bit-equal to `u8 as f32 / 255.0`, labels `u8 as f32`
    (the reference's real label bytes), then a shuffled DataLoader epoch over it is a permutation of the same rows"""
    import taper_amd as T
    lab_bytes = LABELS.read_bytes()
    y_ref = OX.load_idx_labels(lab_bytes)
    n = y_ref.size
    rng = np.random.default_rng(2)
    px = rng.integers(0, 256, (n, 784), dtype=np.uint8)
    px[0, :256] = np.arange(256, dtype=np.uint8)             # every u8 value: the f32 division must round like the reference
    ip = tmp_path / "train_images"
    ip.write_bytes(_images_idx(px))
    x_ref = OX.load_idx_images(ip.read_bytes())
    ds = T.MNISTDataset.from_idx(ip, LABELS)
    assert ds.len() == n
    xt, yt = ds.tensors()
    assert xt.shape() == (n, 784) and yt.shape() == (n,)
    assert np.array_equal(xt.data().reshape(n, 784), x_ref)
    assert np.array_equal(yt.data(), y_ref)
    assert yt.data()[:20].astype(int).tolist() == MNIST_FIRST_LABELS
    # mnist.rs:327-386: batches of a shuffled epoch cover every sample exactly once, last partial batch kept
    loader = T.DataLoader(ds, 300, True, seed=7)
    seen, sizes = [], []
    for xb, yb in loader:
        xb, yb = xb.data().reshape(-1, 784), yb.data()
        sizes.append(len(yb))
        # identify rows by their content: pixel row -> index (rows are random, collisions impossible in practice)
        seen.append((xb, yb))
    assert sizes == [300] * (n // 300) + [n % 300]
    X = np.concatenate([s[0] for s in seen])
    Y = np.concatenate([s[1] for s in seen])
    key = lambda a: [r.tobytes() for r in a]
    order = {k: i for i, k in enumerate(key(x_ref))}
    perm = np.array([order[k] for k in key(X)])
    assert sorted(perm.tolist()) == list(range(n)) and not np.array_equal(perm, np.arange(n))
    assert np.array_equal(Y, y_ref[perm])
