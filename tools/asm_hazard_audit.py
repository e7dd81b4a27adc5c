#!/usr/bin/env python3
"""Static check of the inline-asm vector-memory statements in the library's ISA (hipcc -S): what the compiler cannot see, it does not protect.

  * an asm LOAD's destination registers count as written at the end of the statement: no instruction may read, copy or overwrite them before
    the next `s_waitcnt vmcnt(0)` (cdna_hip_programming.md, "What hipcc does not do", form (ii));
  * an asm STORE of more than 64 bits keeps reading its data registers after it has issued, and the hazard recognizer does not know the
    statement is a store: the store must be followed -- inside the same statement -- by another asm store or by `s_waitcnt vmcnt(0)`, never by
    compiler-scheduled code (r04's k split wrote a wrong step in 5 - 8 of 30 runs on a shared GPU because of this).

usage: asm_hazard_audit.py file.hip [...]   (run from taper_amd/csrc or give paths; exit status 1 on a finding)"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HIPCC = "/opt/rocm/bin/hipcc"


def isa(src: Path) -> list[str]:
    with tempfile.TemporaryDirectory() as d:
        out = Path(d) / "k.s"
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", f"-I{ROOT / 'include'}",
                               f"-I{src.parent}", "-S", "--cuda-device-only", "-o", str(out), str(src)], stderr=subprocess.DEVNULL)
        return out.read_text().split("\n")


def regs_of(text: str) -> set[int]:
    r = {int(x) for x in re.findall(r"\bv(\d+)\b", text)}
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        r |= set(range(int(a), int(b) + 1))
    return r


def audit(lines: list[str]) -> list[str]:
    findings = []
    fn = "?"
    in_asm = False
    i = 0
    while i < len(lines):
        t = lines[i].strip()
        if re.match(r"^_Z\w+:", lines[i]):
            fn = lines[i].split(":")[0]
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif in_asm and re.match(r"(global|buffer|flat)_load_dword(x[234])? ", t) and " lds" not in t:
            dest = regs_of(t.split(",")[0])
            j = i + 1
            while j < len(lines) and "s_waitcnt vmcnt(0)" not in lines[j] and not lines[j].startswith(".Lfunc_end"):
                u = lines[j].strip()
                if u and not u.startswith((";", ".", "s_")) and not re.match(r"(global|buffer|flat)_load", u) and regs_of(u) & dest:
                    findings.append(f"{fn}: `{u}` touches the destination of asm load `{t}` before its wait")
                j += 1
        elif in_asm and re.match(r"(global|buffer|flat)_store_dwordx[34] ", t):
            nxt = lines[i + 1].strip()
            if not (re.match(r"(global|buffer|flat)_store", nxt) or nxt.startswith("s_waitcnt vmcnt(0)")):
                findings.append(f"{fn}: asm store `{t}` is not followed by its wait inside the statement (next: `{nxt}`)")
        i += 1
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
