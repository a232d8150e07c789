import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from bench import make_workload
wl = make_workload(num_cells=(4, 4, 2), frames=10000, hparams="perf", seed=33)
model = wl["model"](device=0)
pos = wl["positions"]
model.calc_polarizabilities(pos[:2000])
for rep in range(3):
    t = time.perf_counter(); a = model.calc_polarizabilities(pos); dt = time.perf_counter() - t
    print("host numpy calc_polarizabilities: %.0f structures/s" % (len(pos) / dt))
dpos = torch.as_tensor(pos, device="cuda")
model.calc_polarizabilities_device(dpos, synchronize=True)
t = time.perf_counter(); b = model.calc_polarizabilities_device(dpos, synchronize=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("device-resident: %.0f structures/s" % (len(pos) / dt), "equal:", np.array_equal(a, b.cpu().numpy()))
