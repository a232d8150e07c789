#!/usr/bin/env python3
"""Time a few training steps (forward + backward + Adam) for profiling."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload, rocksalt
hp = sys.argv[1] if len(sys.argv) > 1 else "perf"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cells = (4, 4, 2) if os.environ.get("RN_PROBE_256") else (4, 2, 2)
wl = make_workload(cells, batch, hp, seed=55)
model = wl["model"]()
lattice, ref, zs = rocksalt(*cells)
lat = torch.tensor(lattice, dtype=torch.float32).expand(batch, 3, 3)
z = torch.tensor(zs).expand(batch, -1)
pos = torch.tensor(wl["positions"], dtype=torch.float32)
target = torch.randn(batch, 6)
if os.environ.get("RN_PROBE_DEVICE"):
    from ramannoodle_amd.pmodel import DeviceAdam
    opt = DeviceAdam(model, lr=1e-3)
else:
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
model.train()
if os.environ.get("RN_PROBE_NOGC"):
    import gc
    gc.disable()
if os.environ.get("RN_PROBE_THREADS"):
    torch.set_num_threads(int(os.environ["RN_PROBE_THREADS"]))
for it in range(int(os.environ.get("RN_PROBE_STEPS", "6"))):
    t0 = time.perf_counter()
    out = model.forward(lat, z, pos)
    t1 = time.perf_counter()
    loss = torch.nn.functional.mse_loss(out, target)
    loss.backward()
    t2 = time.perf_counter()
    opt.step(); opt.zero_grad()
    t3 = time.perf_counter()
    print(f"step {it}: fwd {1e3*(t1-t0):.1f} ms  bwd {1e3*(t2-t1):.1f} ms  opt {1e3*(t3-t2):.1f} ms")
