"""One structure per call through calc_polarizabilities, a few hundred times: for `rocprofv3 --kernel-trace` (where do the
~450 us of a batch-1 call go?).  usage: batch1_trace.py perf|parity [calls]"""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
wl = bench.make_workload(num_cells=(4, 4, 2), frames=64, hparams=sys.argv[1], seed=33)
model = wl["model"](device=0)
pos = wl["positions"]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for i in range(20):
    model.calc_polarizabilities(pos[i % 64][None])
t0 = time.perf_counter()
for i in range(calls):
    model.calc_polarizabilities(pos[i % 64][None])
print("us per call", (time.perf_counter() - t0) / calls * 1e6)
