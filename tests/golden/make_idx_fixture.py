#!/usr/bin/env python3
"""Cuts tests/golden/mnist_train_labels_head.idx1 out of the reference's data file
`/root/reference/data/mnist/train_labels` (the real MNIST training labels in IDX1 format; the image files are
not part of the reference checkout): header magic 0x00000801, count rewritten to N, then the first N label bytes.
A data file is a fixture, not source; the reference never travels to the GPU box, this slice does.
Run (authoring container only):  python tests/golden/make_idx_fixture.py
"""
import struct
from pathlib import Path

N = 2048
src = Path("/root/reference/data/mnist/train_labels").read_bytes()
magic, count = struct.unpack(">II", src[:8])
assert magic == 0x801 and count == 60000 and len(src) == 8 + count
out = Path(__file__).resolve().parent / "mnist_train_labels_head.idx1"
out.write_bytes(struct.pack(">II", 0x801, N) + src[8:8 + N])
print(out, out.stat().st_size, "bytes; first 16 labels:", list(src[8:24]))
