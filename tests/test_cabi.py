"""The C-ABI libraries load WITHOUT a GPU and export every symbol the headers
declare (no compute calls here); the product path has no CPU fallback."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared(header):
    text = (ROOT / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:th|tp)_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("so,header,min_syms", [("libtaper_hip.so", "taper_hip.h", 70), ("libtaper_hip.so", "taper_hip_debug.h", 5),
                                                ("libtaper_host.so", "taper_host.h", 90)])
def test_library_exports_every_declared_symbol(so, header, min_syms):
    path = ROOT / "taper_amd" / "lib" / so
    assert path.exists(), f"{path} missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(str(path), mode=ctypes.RTLD_GLOBAL)
    names = _declared(header)
    assert len(names) >= min_syms
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"{so} does not export: {missing}"


def test_header_parser_matches_declarations():
    from taper_amd import _lib
    assert sorted(_lib.HIP_PROTOS) == sorted(_declared("taper_hip.h") + _declared("taper_hip_debug.h"))
    assert sorted(_lib.HOST_PROTOS) == _declared("taper_host.h")
    # the boundary itself holds no test hooks
    assert not [n for n in _declared("taper_hip.h") if n.startswith("th_debug")]
    # spot-check a few parsed signatures against the header text
    r, a = _lib.HIP_PROTOS["th_sgemm"]
    assert r is ctypes.c_int and len(a) == 11 and a[1] is ctypes.c_int and a[6] is ctypes.c_float and a[7] is ctypes.c_void_p
    r, a = _lib.HIP_PROTOS["th_adam_step"]
    assert len(a) == 16 and a[8] is ctypes.c_int64 and a[11] is ctypes.c_float
    assert _lib.HIP_PROTOS["th_last_error"][0] is ctypes.c_char_p


def test_no_cpu_fallback_without_gpu():
    """On a box without a GPU the product path must fail loudly, never compute on the host."""
    import taper_amd as T
    from taper_amd import hip
    if hip.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(T.TaperError, match="ROCm|device|MI355X"):
        T.Tensor([1.0, 2.0])
    with pytest.raises(T.TaperError):
        hip.Ctx(0)


def test_product_code_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under taper_amd/ may import, link or call it."""
    for p in (ROOT / "taper_amd").rglob("*"):
        if p.suffix in {".py", ".cpp", ".h", ".hip"} or p.name == "Makefile":
            text = p.read_text(errors="ignore")
            assert "oracle" not in text.lower(), f"{p} mentions the oracle"
