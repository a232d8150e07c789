#!/usr/bin/env python3
"""Worst relative error of the device path against every reference fixture (alpha, float32 path)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.conftest import load_golden, CASES
from tests.helpers import product_model_from_golden
for case in CASES:
    g = load_golden(case)
    if "f32/alpha" not in g.files:
        continue
    model = product_model_from_golden(g)
    got = model.calc_polarizabilities(g["pos_batch"])
    mean, std = g["mean"], g["std"]
    line = f"{case:22s} frames {got.shape[0]:4d}"
    for ref in ("f32/alpha", "f64/alpha"):
        want = g[ref]
        rel = np.abs(got - want).max() / np.abs(want).max()
        rel_std = np.abs((got - want) / std).max() / np.abs((want - mean) / std).max()
        line += f"   vs reference {ref[:3]}: {rel:.1e} (standardised {rel_std:.1e})"
    flags = model.config_flags()
    print(line + f"   fused={int(flags['fused_edge_block'])} folded_gate={int(flags['folded_gate_scale'])}"
                 f" split_f16={int(flags['split_f16_mfma'])} narrow={int(flags['narrow_kernels'])}")
