#!/usr/bin/env python3
"""Print a rocprofv3 --stats kernel summary (CSV) as a table; used to make profiles/*.txt."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
total = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':84s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>8s} {'pct':>6s}")
for r in rows:
    if float(r["TotalDurationNs"]) / total < 0.0005:
        continue
    print(f"{r['Name'][:84]:84s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} "
          f"{float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):6.1f}")
print(f"total kernel time {total/1e6:.2f} ms")
