#!/usr/bin/env python3
"""Build-time check of the hand-counted `s_waitcnt vmcnt(N)` waits (kernels_edge_ps.hip, kernels_node_atom.hip).

The role-specialised EdgeBlock's producers issue their node-term loads in inline assembly and wait for them with a hand-counted
`s_waitcnt vmcnt(5)` many instructions later (the compiler cannot be told to wait for "all but the five LDS-DMA requests issued
since").  That is correct only for the instruction stream this hipcc emits, so the build checks the stream (csrc/build.sh runs
this script after compiling and fails on a violation; tests/test_host_logic.py runs it too):

  1. until the wait the destination registers are in flight: no instruction may read, copy or overwrite them;
  2. a counted wait `vmcnt(N)`, N > 0, behind a group of asm loads returns when at most N younger VMEM operations are
     outstanding: between the group's last load and the wait there must be EXACTLY N `global_load_lds` requests and no
     other VMEM operation (a load, a store, scratch traffic) -- fewer would let the wait return with node terms in flight,
     more would only make it stricter but means the compiler added traffic (a spill) nobody asked for.  The one exception is
     the failure path of a bounded spin wait (a `global_store` followed at once by `s_waitcnt vmcnt(0)`: it drains);
  3. no instantiation of the EdgeBlock kernel spills a VGPR or uses scratch, and the NodeBlock kernel has no scratch
     traffic inside its round loop (between its counted waits);
  4. the atom-owning NodeBlock (kernels_node_atom.hip) waits with `vmcnt(1)` for the operand rows requested three rounds ago:
     every round must issue its one LDS-DMA request between two such waits (a round without it would let the next wait
     return early).

    python3 tools/check_ps_isa.py            # compiles both files with -save-temps and checks them
    python3 tools/check_ps_isa.py FILE.s ... # checks existing assembly listings
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VMEM = re.compile(r"^(global_|buffer_|scratch_|flat_)")


def regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def kernels(lines):
    """(name, first line, last line) of every kernel body in the listing."""
    out, name, first = [], None, 0
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m and name is None and i + 1 < len(lines):
            name, first = m.group(1), i
        if ln.strip().startswith(".end_amdhsa_kernel") and name is not None:
            out.append((name, first, i))
            name = None
    return out


def instructions(lines, first, last):
    """(line number, text, inside an asm block) of every instruction of a kernel body, in layout order."""
    out, in_asm = [], False
    for i in range(first, last):
        s = lines[i].strip()
        if "#ASMSTART" in s:
            in_asm = True
            continue
        if "#ASMEND" in s:
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":") or s.startswith("//"):
            continue
        out.append((i + 1, s, in_asm))
    return out


def check_edge_ps(path, name, ins):
    bad = checked = counted = 0
    pending = []  # (line, registers) of asm loads not yet waited for
    since = []    # VMEM instructions since the last asm load: (index in ins, text)
    for k, (ln, s, in_asm) in enumerate(ins):
        if in_asm and s.startswith("global_load_dwordx4"):
            pending.append((ln, regs(s.split(",")[0])))
            since = []
            checked += 1
            continue
        if in_asm and s.startswith("s_waitcnt") and "vmcnt" in s:
            m = re.search(r"vmcnt\((\d+)\)", s)
            n = int(m.group(1)) if m else 0
            # the failure path of a bounded spin wait (a cold block laid out in between: store the code, drain, go on) is
            # not on the path from the loads to their wait
            if n == 0 and "lgkmcnt(0)" in s and any(t.startswith("global_store_dword") for _, t, _ in ins[max(0, k - 4):k]):
                continue
            if pending and n > 0:
                counted += 1
                dma = [t for _, t in since if t.startswith("global_load_lds")]
                other = [(j, t) for j, t in since if not t.startswith("global_load_lds")]
                if len(dma) != n:
                    print(f"{os.path.basename(path)}:{ln}: `{s}` behind asm loads with {len(dma)} LDS-DMA requests in between, not {n} ({name})")
                    bad += 1
                for j, t in other:
                    drained = any(x[2] and x[1].startswith("s_waitcnt") and "vmcnt(0)" in x[1] for x in ins[j + 1:j + 9])
                    if not drained:
                        print(f"{os.path.basename(path)}:{ins[j][0]}: `{t}` between asm loads and their counted wait at line {ln} ({name})")
                        bad += 1
            pending, since = [], []
            continue
        if VMEM.match(s) and pending:
            since.append((k, s))
        if in_asm:
            continue
        touched = regs(s)
        for at, dst in pending:
            if touched & dst:
                print(f"{os.path.basename(path)}:{ln}: {s}\n    touches v{sorted(touched & dst)} loaded at line {at} and not waited for yet ({name})")
                bad += 1
    return checked, counted, bad


def check_node_atom(path, name, ins):
    bad = waits = 0
    dma_since = None  # LDS-DMA requests since the previous asm vmcnt(1)
    counted = [k for k, (_, s, a) in enumerate(ins) if a and s.startswith("s_waitcnt") and re.search(r"vmcnt\(1\)", s)]
    for k, (ln, s, in_asm) in enumerate(ins):
        if s.startswith("scratch_") and counted and counted[0] < k < counted[-1]:
            print(f"{os.path.basename(path)}:{ln}: `{s}` inside the round loop ({name})")
            bad += 1
    for ln, s, in_asm in ins:
        if s.startswith("global_load_lds") and dma_since is not None:
            dma_since += 1
        if in_asm and s.startswith("s_waitcnt") and re.search(r"vmcnt\(1\)", s):
            waits += 1
            if dma_since is not None and dma_since < 1:
                print(f"{os.path.basename(path)}:{ln}: `{s}` with no LDS-DMA request since the previous counted wait ({name})")
                bad += 1
            dma_since = 0
    return waits, bad


def resources(lines):
    """{kernel symbol: (vgpr spills, scratch bytes)} from the .amdhsa metadata."""
    out, sym = {}, None
    for ln in lines:
        m = re.match(r"\s+\.name:\s+(\S+)", ln)
        if m:
            sym = m.group(1)
            out.setdefault(sym, [0, 0])
        m = re.match(r"\s+\.vgpr_spill_count:\s+(\d+)", ln)
        if m and sym:
            out[sym][0] = int(m.group(1))
        m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", ln)
        if m and sym:
            out[sym][1] = int(m.group(1))
    return out


def check(path):
    lines = open(path).read().split("\n")
    bad = 0
    res = resources(lines)
    for name, first, last in kernels(lines):
        is_ps, is_na = "edge_block_ps_kernel" in name, "node_block_atom_kernel" in name
        if not (is_ps or is_na):
            continue
        spills, scratch = res.get(name, (0, 0))
        if is_ps and (spills or scratch):
            print(f"{os.path.basename(path)}: {name} spills {spills} VGPRs / uses {scratch} bytes of scratch")
            bad += 1
        ins = instructions(lines, first, last)
        if is_ps:
            checked, counted, b = check_edge_ps(path, name, ins)
            print(f"check_ps_isa: {name}: {checked} in-flight loads, {counted} counted wait(s), {b} violation(s)")
            if checked == 0 or counted == 0:
                print(f"{os.path.basename(path)}: {name}: found no asm loads / counted waits to check -- the check itself is stale")
                b += 1
        else:
            waits, b = check_node_atom(path, name, ins)
            print(f"check_ps_isa: {name}: {waits} counted wait(s), {b} violation(s)")
            if waits == 0:
                print(f"{os.path.basename(path)}: {name}: found no counted waits to check -- the check itself is stale")
                b += 1
        bad += b
    return bad


def compile_and_check(names):
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for name in names:
            src = os.path.join(ROOT, "ramannoodle_amd", "csrc", name + ".hip")
            cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
                   "-save-temps=obj", "-c", src, "-o", os.path.join(tmp, name + ".o")] + os.environ.get("RN_EXTRA_FLAGS", "").split()
            subprocess.check_call(cmd, cwd=os.path.dirname(src))
            bad += check(os.path.join(tmp, name + "-hip-amdgcn-amd-amdhsa-gfx950.s"))
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 1:
        total = sum(check(p) for p in sys.argv[1:])
    else:
        total = compile_and_check(["kernels_edge_ps", "kernels_node_atom"])
    print(f"check_ps_isa: {total} violation(s) in all")
    sys.exit(1 if total else 0)
