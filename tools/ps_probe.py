"""Prints which EdgeBlock kernel a bench-shaped model selects and what a launch costs (GPU box)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from bench import make_workload

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
wl = make_workload(num_cells=(4, 4, 2), frames=frames, hparams="perf", seed=33)
model = wl["model"](device=0)
print("flags", model.config_flags())
pos = torch.as_tensor(wl["positions"], device="cuda")
model.set_profiling(1)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model.calc_polarizabilities_device(pos)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("rep", rep, "%.1f structures/s" % (frames / dt))
for k, v in model.kernel_times().items():
    print("%-16s %9.3f ms %5d launches" % (k, v[0], v[1]))

# a synchronising host-buffer call: timing builds (RN_PS_TIMING) print their phase counters to stderr here
_ = model.calc_polarizabilities(wl["positions"][: min(frames, 2000)])
