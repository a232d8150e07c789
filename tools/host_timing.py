"""Where a `calc_polarizabilities(numpy)` call of config 3's 8-GPU share (1250 frames) and of one structure spends its time,
next to the HBM-resident entry (RN_POTGNN_HOST_TIMING=1 prints the C side's split).  usage: host_timing.py perf|parity"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, bench
wl = bench.make_workload(num_cells=(4, 4, 2), frames=1250, hparams=sys.argv[1], seed=33)
model = wl["model"](device=0)
pos = wl["positions"]
def best(f, n=5):
    f(); r = []
    for _ in range(n):
        t = time.perf_counter(); f(); r.append(time.perf_counter() - t)
    return min(r)
d = torch.tensor(pos, device="cuda"); out = torch.empty((1250, 3, 3), dtype=torch.float64, device="cuda")
h = best(lambda: model.calc_polarizabilities(pos)); r = best(lambda: model.calc_polarizabilities_device(d, out, synchronize=True))
print("1250 frames: host %.3f ms  resident %.3f ms  ratio %.3f" % (h * 1e3, r * 1e3, r / h))
h = best(lambda: model.calc_polarizabilities(pos[:1]), 50); r = best(lambda: model.calc_polarizabilities_device(d[:1], out[:1], synchronize=True), 50)
print("1 frame: host %.0f us  resident %.0f us" % (h * 1e6, r * 1e6))
