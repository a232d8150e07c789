#!/usr/bin/env python3
"""Throughput (structures/s, inputs resident in HBM) of the 128-atom workload for a list of embedding
widths, plus the largest deviation from the CPU oracle on 2 frames:
  python3 tools/width_sweep.py 32x32 24x24 32x64 64x32 20x48 [frames]
With RN_POTGNN_WIDEN=1 the widths 17..64 are padded to 64 x 64 (fused MFMA kernels)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

frames = 2000
cases = []
for a in sys.argv[1:]:
    if "x" in a:
        fn, fe = a.split("x")
        cases.append((int(fn), int(fe)))
    else:
        frames = int(a)
for fn, fe in cases:
    name = f"w{fn}x{fe}"
    bench.HPARAMS[name] = (fn, fe, 4)
    wl = bench.make_workload((4, 2, 2), frames, name, seed=91)
    model = wl["model"]()
    pos = torch.tensor(wl["positions"], device="cuda")
    model.calc_polarizabilities_device(pos, synchronize=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        out = model.calc_polarizabilities_device(pos, synchronize=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    from oracle import potgnn_oracle as O
    ref = np.asarray(O.calc_polarizabilities(wl["oracle"](), wl["positions"][:2], faithful=False))
    err = float(np.abs(out[:2].cpu().numpy() - ref).max() / np.abs(ref).max())
    flags = model.config_flags()
    path = "narrow" if flags.get("narrow_kernels") else ("fused" if flags.get("fused_edge_block") else "unfused")
    print(f"Fn={fn:3d} Fe={fe:3d}  {frames / dt:10.0f} structures/s  {1e6 * dt / frames:7.2f} us/structure  "
          f"path={path}  max rel dev vs oracle {err:.2e}", flush=True)
