"""Thin numpy <-> device helpers over the raw kernel ABI (include/taper_hip.h).

Used by the per-kernel parity tests and bench.py: `ctx.call("th_sgemm", ...)`
goes straight through the C ABI, nothing is computed on the host.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import HIP_PROTOS, hip, th_check


class AdamFuse(C.Structure):
    """include/taper_hip.h: th_adam_fuse (device pointers as integers)"""
    _fields_ = [("d_p", C.c_void_p), ("d_m", C.c_void_p), ("d_v", C.c_void_p), ("d_t", C.c_void_p), ("d_lr", C.c_void_p),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float)]


class AdamSlice(C.Structure):
    """include/taper_hip.h: th_adam_slice"""
    _fields_ = [("d_g", C.c_void_p), ("n", C.c_int64), ("f", AdamFuse)]


class WideFuse(C.Structure):
    """include/taper_hip.h: th_wide_fuse"""
    _fields_ = [("w", AdamFuse), ("b", AdamFuse), ("conv_b", AdamFuse), ("d_conv_gb", C.c_void_p), ("conv_c", C.c_int), ("conv_hw", C.c_int)]


class Mlp3Layer(C.Structure):
    """include/taper_hip.h: th_mlp3_layer"""
    _fields_ = [("d_w", C.c_void_p), ("d_b", C.c_void_p), ("d_dw", C.c_void_p), ("d_db", C.c_void_p), ("w_fuse", C.c_void_p),
                ("b_fuse", C.c_void_p), ("out_features", C.c_int)]


class Mlp3Gap(C.Structure):
    """include/taper_hip.h: th_mlp3_gap"""
    _fields_ = [("d_cnt", C.c_void_p), ("d_gb", C.c_void_p), ("hw", C.c_int), ("b_fuse", C.c_void_p)]


class RowSource(C.Structure):
    """include/taper_hip.h: th_row_source"""
    _fields_ = [("d_rows", C.c_void_p), ("d_labels", C.c_void_p), ("d_indices", C.c_void_p), ("d_cursor", C.c_void_p),
                ("n_indices", C.c_int64), ("n_rows", C.c_int64)]


class ConvStage(C.Structure):
    """include/taper_hip.h: th_conv_stage"""
    _fields_ = [("d_w", C.c_void_p), ("d_bias", C.c_void_p), ("c_out", C.c_int), ("post", C.c_int)]


CHAIN_NONE, CHAIN_MAXPOOL2, CHAIN_GLOBAL_AVG = 0, 1, 2


def conv_stages(stages):
    """[(w DevBuf, bias DevBuf, c_out, post), ...] -> (ctypes array of th_conv_stage, count)"""
    arr = (ConvStage * len(stages))()
    for i, (w, b, c_out, post) in enumerate(stages):
        arr[i] = ConvStage(int(w), int(b), int(c_out), int(post))
    return arr, len(stages)


class DevBuf:
    """A device allocation from the ctx pool (freed on garbage collection)."""

    def __init__(self, ctx: "Ctx", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = C.c_void_p()
        th_check(hip.th_malloc(ctx.h, max(self.nbytes, 4), C.byref(p)), "th_malloc")
        self.ptr = p.value

    def __del__(self):
        if getattr(self, "ptr", None) and self.ctx.h:
            hip.th_free(self.ctx.h, self.ptr)
            self.ptr = None

    def __int__(self):
        return self.ptr

    def offset(self, nbytes: int) -> int:
        return self.ptr + int(nbytes)


class Event:
    def __init__(self):
        p = C.c_void_p()
        th_check(hip.th_event_create(C.byref(p)), "th_event_create")
        self.h = p.value

    def __del__(self):
        if getattr(self, "h", None):
            hip.th_event_destroy(self.h)
            self.h = None


class Ctx:
    """One GPU + one HIP stream + the pooled allocator (th_ctx)."""

    def __init__(self, device: int = 0, handle: int | None = None):
        self._owned = handle is None
        if handle is None:
            p = C.c_void_p()
            th_check(hip.th_ctx_create(int(device), C.byref(p)), "th_ctx_create")
            handle = p.value
        self.h = handle

    def close(self):
        if self.h and self._owned:
            hip.th_ctx_destroy(self.h)
        self.h = None

    # -- memory -----------------------------------------------------------
    def empty(self, n: int, dtype=np.float32) -> DevBuf:
        return DevBuf(self, int(n) * np.dtype(dtype).itemsize)

    def upload(self, a) -> DevBuf:
        a = np.ascontiguousarray(a)
        b = DevBuf(self, a.nbytes)
        th_check(hip.th_memcpy_h2d(self.h, b.ptr, a.ctypes.data, a.nbytes), "th_memcpy_h2d")
        return b

    def download(self, buf, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        th_check(hip.th_memcpy_d2h(self.h, out.ctypes.data, int(buf), out.nbytes), "th_memcpy_d2h")
        return out

    def zeros(self, n: int) -> DevBuf:
        b = self.empty(n)
        self.call("th_fill_f32", b, 0.0, n)
        return b

    def sync(self):
        th_check(hip.th_ctx_sync(self.h), "th_ctx_sync")

    # -- ops --------------------------------------------------------------
    def call(self, name: str, *args):
        """name(ctx, *args) through the C ABI; DevBuf / None / ints are accepted for pointers."""
        if name not in HIP_PROTOS:
            raise AttributeError(f"{name} is not declared in include/taper_hip.h (or taper_hip_debug.h)")
        conv = [int(a) if isinstance(a, DevBuf) else a for a in args]
        th_check(getattr(hip, name)(self.h, *conv), name)

    # -- timing / graphs ------------------------------------------------------
    def record(self, ev: Event):
        th_check(hip.th_event_record(self.h, ev.h), "th_event_record")

    @staticmethod
    def elapsed_ms(start: Event, stop: Event) -> float:
        ms = C.c_float()
        th_check(hip.th_event_elapsed_ms(start.h, stop.h, C.byref(ms)), "th_event_elapsed_ms")
        return float(ms.value)

    def graph_begin(self):
        th_check(hip.th_graph_begin(self.h), "th_graph_begin")

    def graph_end(self) -> int:
        g = C.c_void_p()
        th_check(hip.th_graph_end(self.h, C.byref(g)), "th_graph_end")
        return g.value

    def graph_launch(self, g: int):
        th_check(hip.th_graph_launch(self.h, g), "th_graph_launch")

    @staticmethod
    def graph_destroy(g: int):
        hip.th_graph_destroy(g)


def device_count() -> int:
    n = C.c_int(0)
    rc = hip.th_device_count(C.byref(n))
    return n.value if rc == 0 else 0
