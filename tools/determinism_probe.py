"""Evaluate one trajectory several times and count frames whose result differs between runs
(the kernels have no atomics: any difference is a race).  usage: determinism_probe.py [frames] [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
wl = make_workload((4, 4, 2), frames, "perf", seed=33)
VARIANTS = ({}, {"RN_POTGNN_MFMA": "f32"}, {"RN_POTGNN_NODE_FUSED": "0"}, {"RN_POTGNN_READOUT_FUSED": "0"},
            {"RN_POTGNN_FUSED": "0"})
if os.environ.get("RN_PROBE_DEFAULT_ONLY"):
    VARIANTS = ({},)
for env in VARIANTS:
    for k in ("RN_POTGNN_MFMA", "RN_POTGNN_NODE_FUSED", "RN_POTGNN_READOUT_FUSED", "RN_POTGNN_FUSED"):
        os.environ.pop(k, None)
    os.environ.update(env)
    model = wl["model"]()
    ref = model.calc_polarizabilities(wl["positions"])
    bad = []
    for _ in range(reps):
        out = model.calc_polarizabilities(wl["positions"])
        diff = np.abs(out - ref).reshape(frames, -1).max(axis=1)
        bad.append(int((diff > 0).sum()))
        worst = diff.max() / np.abs(ref).max()
    print(env or "default", "frames differing from the first run:", bad, f"worst rel {worst:.1e}", flush=True)
    del model
