#!/usr/bin/env python3
"""Build-time check of the role-specialised EdgeBlock's producer loop (kernels_edge_ps.hip).

The producers issue their node-term loads in inline assembly and wait for them with a hand-counted `s_waitcnt vmcnt(N)` many
instructions later (the compiler cannot be told to wait for "all but the five LDS-DMA requests issued since").  Until that
wait the destination registers are in flight: nothing may read, copy or spill them.  This script compiles the file with
-save-temps, finds every `global_load_dwordx4` that came from an asm block in every instantiation of the kernel, and checks that
no instruction touches its destination registers before the next asm `s_waitcnt vmcnt`.  Exit status 1 on a violation.

    python3 tools/check_ps_isa.py [path/to/kernels_edge_ps-hip-amdgcn-amd-amdhsa-gfx950.s]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(path):
    lines = open(path).read().split("\n")
    bad = 0
    checked = 0
    kernel = None
    in_asm = False
    pending = []  # (line number, destination registers) of asm loads not yet waited for
    for i, ln in enumerate(lines, 1):
        s = ln.strip()
        m = re.match(r"^(_ZN2rn20edge_block_ps_kernel\w+):", ln)
        if m:
            kernel, pending = m.group(1), []
        if s.startswith(".end_amdhsa_kernel"):
            kernel = None
        if kernel is None or not s or s.startswith(";") and "ASM" not in s:
            continue
        if "#ASMSTART" in s:
            in_asm = True
            continue
        if "#ASMEND" in s:
            in_asm = False
            continue
        if in_asm and s.startswith("global_load_dwordx4"):
            dst = regs(s.split(",")[0])
            pending.append((i, dst))
            checked += 1
            continue
        if in_asm and s.startswith("s_waitcnt vmcnt"):
            pending = []
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        touched = regs(s)
        for at, dst in pending:
            if touched & dst:
                print(f"{os.path.basename(path)}:{i}: {s}\n    touches v{sorted(touched & dst)} loaded at line {at} and not waited for yet ({kernel})")
                bad += 1
    print(f"check_ps_isa: {checked} in-flight loads checked, {bad} violation(s)")
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1:
        sys.exit(1 if check(sys.argv[1]) else 0)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(ROOT, "ramannoodle_amd", "csrc", "kernels_edge_ps.hip")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
               "-save-temps=obj", "-c", src, "-o", os.path.join(tmp, "k.o")] + os.environ.get("RN_EXTRA_FLAGS", "").split()
        subprocess.check_call(cmd, cwd=os.path.dirname(src))
        sys.exit(1 if check(os.path.join(tmp, "kernels_edge_ps-hip-amdgcn-amd-amdhsa-gfx950.s")) else 0)
