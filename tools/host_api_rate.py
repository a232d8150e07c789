"""The host boundary where callers use it: `PotGNN.calc_polarizabilities(numpy)` on the whole trajectory, on config 3's
8-GPU share and one structure per call, each beside the HBM-resident rate of the same frames (bench.py host_boundary).
usage: python tools/host_api_rate.py [perf|parity|tio2]"""
import json
import sys
sys.path.insert(0, ".")
import bench

which = sys.argv[1] if len(sys.argv) > 1 else "perf"
if which == "tio2":
    wl = bench.make_workload(frames=20_000, hparams="parity", seed=108, structure=bench.tio2_structure(), cutoff=2.0)
else:
    wl = bench.make_workload(num_cells=(4, 4, 2), frames=10000, hparams=which, seed=33)
model = wl["model"](device=0)
print(which, json.dumps(bench.host_boundary(model, wl["positions"]), indent=1))
