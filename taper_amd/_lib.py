"""Loads the two native libraries and declares their ctypes signatures by
parsing the C-ABI headers (include/taper_hip.h + taper_hip_debug.h, include/taper_host.h), so the
Python binding can never drift from the boundary a Rust host would bind.

There is NO CPU fallback: if the libraries are missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import re
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
INCLUDE = ROOT / "include"
LIBDIR = Path(__file__).resolve().parent / "lib"
CSRC = Path(__file__).resolve().parent / "csrc"

_SCALARS = {
    "int": C.c_int, "float": C.c_float, "size_t": C.c_size_t, "int64_t": C.c_int64, "uint64_t": C.c_uint64,
    "int32_t": C.c_int32, "uint8_t": C.c_uint8, "double": C.c_double,
}


def _ctype(decl: str):
    decl = decl.strip()
    if "*" in decl or "[" in decl:
        return C.c_void_p
    toks = [t for t in decl.replace("const", " ").split() if t]
    # last token is the parameter name unless the declaration is a bare type
    ty = toks[0] if len(toks) >= 1 else "int"
    if ty not in _SCALARS:
        raise ValueError(f"unhandled C type in header: {decl!r}")
    return _SCALARS[ty]


def parse_header(path: Path):
    """-> {name: (restype, [argtypes])} for every `ret name(args);` prototype."""
    text = path.read_text()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"#[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b((?:th|tp)_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        if "*" in ret:
            restype = C.c_char_p if "char" in ret else C.c_void_p
        elif ret.split()[-1] == "void":
            restype = None
        else:
            restype = _SCALARS[ret.split()[-1]]
        argtypes = [] if args in ("", "void") else [_ctype(a) for a in args.split(",")]
        protos[name] = (restype, argtypes)
    return protos


def build_native(verbose: bool = False) -> None:
    """hipcc (gfx950) + g++ builds, in-tree; cross-compiles without a GPU."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", str(CSRC), "-j8"], stdout=out)
    subprocess.check_call(["make", "-C", str(CSRC / "host")], stdout=out)


def _load(name: str, *headers: str):
    so = LIBDIR / name
    if not so.exists():
        raise ImportError(
            f"{so} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "taper_amd has no CPU fallback.")
    lib = C.CDLL(str(so), mode=C.RTLD_GLOBAL)
    protos = {}
    for header in headers:
        protos.update(parse_header(INCLUDE / header))
    for fn, (restype, argtypes) in protos.items():
        try:
            f = getattr(lib, fn)
        except AttributeError as e:
            raise ImportError(f"{so} does not export {fn} declared in {' / '.join(headers)}") from e
        f.restype = restype
        f.argtypes = argtypes
    return lib, protos


# (taper_hip_debug.h: the test hooks, declared apart from the boundary)
hip, HIP_PROTOS = _load("libtaper_hip.so", "taper_hip.h", "taper_hip_debug.h")
host, HOST_PROTOS = _load("libtaper_host.so", "taper_host.h")


class TaperError(RuntimeError):
    pass


def th_check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise TaperError(f"{what}: {hip.th_last_error().decode()}")


def tp_check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise TaperError(f"{what}: {host.tp_last_error().decode()}")
