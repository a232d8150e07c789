#!/usr/bin/env python3
"""Summarise SQ PMC passes per kernel: VALU/MFMA busy share, wait shares, clock.
usage: sq_summary.py sq1_counter_collection.csv [sq2_counter_collection.csv]"""
import csv, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
passes = collections.defaultdict(set)   # counter -> files it was collected in (averaged over)
dur = collections.defaultdict(float); calls = collections.Counter()
seen = set()
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void rn::', '')[:44]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        passes[r['Counter_Name']].add(f)
        key = (f, r['Dispatch_Id'])
        if key not in seen and f == sys.argv[1]:
            seen.add(key); dur[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])); calls[k] += 1
print(f"{'kernel':44s} {'calls':>5s} {'ms':>8s} {'GHz':>5s} {'valu_busy':>9s} {'mfma_busy':>9s} {'wait_any':>8s} {'wait_inst':>9s} {'cyc/valu':>8s} {'waves/simd':>10s}")
for k in sorted(dur, key=lambda x: -dur[x]):
    a = {c: v / max(len(passes[c]), 1) for c, v in acc[k].items()}; t = dur[k] * 1e-9
    if t <= 0 or not a.get('SQ_WAVE_CYCLES'): continue
    clk = a['GRBM_GUI_ACTIVE'] / 8 / t            # cycles/s
    simd_cycles = clk * t * 1024                  # total SIMD-cycles available
    valu = a['SQ_ACTIVE_INST_VALU'] * 4 / simd_cycles
    mfma = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd_cycles
    wc = a['SQ_WAVE_CYCLES']
    print(f"{k:44s} {calls[k]:5d} {t*1e3:8.2f} {clk/1e9:5.2f} {valu:9.2f} {mfma:9.2f} {a['SQ_WAIT_ANY']/wc:8.2f} {a['SQ_WAIT_INST_ANY']/wc:9.2f} {a['SQ_ACTIVE_INST_VALU']*4/max(a['SQ_INSTS_VALU'],1):8.2f} {wc*4/simd_cycles:10.2f}")
