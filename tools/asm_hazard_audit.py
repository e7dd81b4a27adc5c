#!/usr/bin/env python3
"""Static check of the inline-asm vector-memory statements in the library's ISA (hipcc -S): what the compiler cannot see, it does not protect.

  * an asm LOAD's destination registers count as written at the end of the statement: no instruction may read, copy or overwrite them before
    an `s_waitcnt vmcnt(0)` ON ANY PATH from the load (cdna_hip_programming.md, "What hipcc does not do", form (ii)).  The listing's control
    flow is followed -- labels, `s_branch`, `s_cbranch_*`: a wait that sits in another basic block only counts on the paths that pass it,
    and a path that reaches a touching instruction without one is a finding, however the blocks are laid out in the file;
  * an asm STORE (any width) keeps reading its data registers after it has issued, and the hazard recognizer does not know the statement
    is a store: the store must be followed -- inside the same statement -- by another asm store or by `s_waitcnt vmcnt(0)`, never by
    compiler-scheduled code (r04's k split wrote a wrong step in 5 - 8 of 30 runs on a shared GPU because of this).

usage: asm_hazard_audit.py file.hip [...]   (run from taper_amd/csrc or give paths; exit status 1 on a finding)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def hipcc() -> str:
    """$HIPCC, else hipcc on PATH, else /opt/rocm/bin/hipcc"""
    return os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def have_hipcc() -> bool:
    h = hipcc()
    return bool(shutil.which(h) or Path(h).exists())


def isa(src: Path) -> list[str]:
    with tempfile.TemporaryDirectory() as d:
        out = Path(d) / "k.s"
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT / 'include'}",
                               f"-I{src.parent}", "-S", "--cuda-device-only", "-o", str(out), str(src)], stderr=subprocess.DEVNULL)
        return out.read_text().split("\n")


def regs_of(text: str) -> set[int]:
    r = {int(x) for x in re.findall(r"\bv(\d+)\b", text)}
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        r |= set(range(int(a), int(b) + 1))
    return r


LOAD = re.compile(r"(global|buffer|flat)_load_(dword(x[234])?|ubyte|sbyte|ushort|sshort|short_d16\w*) ")
STORE = re.compile(r"(global|buffer|flat)_store_(dword(x[234])?|byte|short)\w* ")
LABEL = re.compile(r"^([.\w$]+):")
WAIT0 = re.compile(r"s_waitcnt\b.*\bvmcnt\(0\)")


def _is_code(u: str) -> bool:
    return bool(u) and not u.startswith((";", ".", "//"))


def _successors(lines, labels, j):
    """indices control may reach after executing line j (a code line)"""
    u = lines[j].strip()
    if u.startswith(("s_endpgm", "s_setpc_b64", "s_swappc_b64")):
        return []
    m = re.match(r"s_branch\s+([.\w$]+)", u)
    if m:
        return [labels[m.group(1)]] if m.group(1) in labels else []
    m = re.match(r"s_cbranch_\w+\s+([.\w$]+)", u)
    if m:
        return [j + 1] + ([labels[m.group(1)]] if m.group(1) in labels else [])
    return [j + 1]


def _walk_load(lines, labels, i, fn_end, t, dest, fn, findings):
    """every path from the asm load at line i until a vmcnt(0) wait: nothing may touch `dest`"""
    seen, stack, reported = set(), [i + 1], set()
    while stack:
        j = stack.pop()
        while j < fn_end:
            if j in seen:
                break
            seen.add(j)
            u = lines[j].strip()
            if not _is_code(u) or LABEL.match(lines[j]):
                j += 1
                continue
            if WAIT0.search(u):
                break
            if not u.startswith("s_") and not LOAD.match(u) and regs_of(u) & dest and j not in reported:
                reported.add(j)
                findings.append(f"{fn}: `{u}` touches the destination of asm load `{t}` before its wait")
            nxt = _successors(lines, labels, j)
            if not nxt:
                break
            stack.extend(nxt[1:])
            j = nxt[0]


def audit(lines: list[str]) -> list[str]:
    findings = []
    labels = {}
    for k, line in enumerate(lines):
        m = LABEL.match(line)
        if m:
            labels[m.group(1)] = k
    fn, fn_end = "?", len(lines)
    in_asm = False
    for i, line in enumerate(lines):
        t = line.strip()
        if re.match(r"^_Z\w+:", line):
            fn = line.split(":")[0]
            fn_end = next((k for k in range(i + 1, len(lines)) if lines[k].startswith(".Lfunc_end")), len(lines))
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and LOAD.match(t) and " lds" not in t:
            _walk_load(lines, labels, i, fn_end, t, regs_of(t.split(",")[0]), fn, findings)
        elif in_asm and STORE.match(t):
            nxt = lines[i + 1].strip() if i + 1 < len(lines) else ""
            if not (STORE.match(nxt) or WAIT0.search(nxt)):
                findings.append(f"{fn}: asm store `{t}` is not followed by its wait inside the statement (next: `{nxt}`)")
    return findings


def main():
    srcs = [Path(a) for a in sys.argv[1:]] or sorted((ROOT / "taper_amd" / "csrc").glob("*.hip"))
    bad = 0
    for s in srcs:
        if "asm volatile" not in s.read_text():
            continue
        f = audit(isa(s))
        print(f"{s.name}: {len(f)} finding(s)")
        for x in f:
            print("   ", x)
        bad += len(f)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
