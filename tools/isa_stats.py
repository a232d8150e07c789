#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps assembly listing, whole body and per innermost loop.

usage: isa_stats.py LISTING.s KERNEL_SUBSTRING [--loops]
A "loop" is the span between a label and the last backward branch to it; only spans that contain no other
backward-branch target are reported (innermost loops).  Issue cycles use the guide's prices: 8 for the
transcendentals, 4 for everything else that issues on the VALU (MI355X_MICROARCH.md, constants table)."""
import collections
import re
import sys

TRANS = ("v_exp_", "v_rcp_", "v_rsq_", "v_log_", "v_sqrt_", "v_sin_", "v_cos_")


def kernel_bodies(text):
    out = {}
    for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)^\s*s_endpgm", text, re.S | re.M):
        out[m.group(1)] = m.group(2)
    return out


def classify(op):
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def stats(lines):
    c = collections.Counter()
    ops = collections.Counter()
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        c[classify(op)] += 1
        ops[op] += 1
    return c, ops


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2]
    for name, body in kernel_bodies(text).items():
        if want not in name:
            continue
        lines = body.split("\n")
        c, ops = stats(lines)
        meta = {}
        for key in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "NumAgprs"):
            m = re.search(r"; %s: (\d+)" % key, text[text.index(name + ":"):])
            if m:
                meta[key] = int(m.group(1))
        print(name)
        print("  ", meta)
        print("   whole body:", dict(c))
        if "--loops" not in sys.argv:
            continue
        labels = {}
        for i, ln in enumerate(lines):
            m = re.match(r"^(\.LBB\w+):", ln)
            if m:
                labels[m.group(1)] = i
        spans = []
        for i, ln in enumerate(lines):
            m = re.search(r"\bs_cbranch_\w+\s+(\.LBB\w+)|\bs_branch\s+(\.LBB\w+)", ln)
            if m:
                tgt = m.group(1) or m.group(2)
                if tgt in labels and labels[tgt] < i:
                    spans.append((labels[tgt], i, tgt))
        inner = [s for s in spans if not any(o is not s and s[0] <= o[0] and o[1] <= s[1] and (o[0], o[1]) != (s[0], s[1]) for o in spans)]
        for a, b, tgt in inner:
            c, ops = stats(lines[a:b + 1])
            n = sum(c.values())
            if n < 40:
                continue
            cyc = 8 * c["trans"] + 4 * (c["v_pk"] + c["valu"] + c["lds"] + c["vmem"]) + 16 * c["mfma"]
            print(f"   loop {tgt}: {n} instr, ~{cyc} issue cycles: {dict(c)}")
            if "--ops" in sys.argv:
                print("      ", ops.most_common(30))


if __name__ == "__main__":
    main()
