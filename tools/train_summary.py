#!/usr/bin/env python3
"""Kernel table of a rocprofv3 --kernel-trace run of tools/train_probe.py (per-kernel calls, total ms, average us,
share), followed by the probe's own per-step wall-clock lines:
  python3 tools/train_summary.py <results.db> <probe stdout log>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name,count(*),sum(d.end-d.start)/1e6,avg(d.end-d.start)/1e3 from {kd} d "
                  f"join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc").fetchall()
total = sum(r[2] for r in rows)


def demangle(name):
    name = re.sub(r"\.kd$", "", name)
    m = re.match(r"_ZN2rn\d+([A-Za-z0-9_]+?)I(.*?)EEv", name)
    if m:
        args = re.sub(r"Li(\d+)E", r"\1,", m.group(2)).replace("Lb1E", "true,").replace("Lb0E", "false,")
        args = args.replace("f", "float,", 1) if re.fullmatch(r"(\d+,)*f(\d+,)*", args) else args
        return f"rn::{m.group(1)}<{args.rstrip(',')}>"
    m = re.match(r"_ZN2rn\d+([A-Za-z0-9_]+?)E", name)
    return f"rn::{m.group(1)}" if m else name


print(f"{'kernel':86s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>8s} {'pct':>6s}")
for name, calls, ms, avg in rows:
    if ms / total < 0.001:
        continue
    print(f"{demangle(name)[:86]:86s} {calls:6d} {ms:9.2f} {avg:8.1f} {100 * ms / total:6.1f}")
print(f"total kernel time {total:.2f} ms")
print()
for line in open(sys.argv[2]):
    if line.startswith("step"):
        print(line.rstrip())
