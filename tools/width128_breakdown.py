import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import bench
for fn, fe in [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or ((128,128),(64,128),(128,64)):
    name=f"w{fn}x{fe}"; bench.HPARAMS[name]=(fn,fe,4)
    wl = bench.make_workload((4,2,2), 1000, name, seed=91)
    model = wl["model"](); pos = torch.tensor(wl["positions"], device="cuda")
    model.calc_polarizabilities_device(pos, synchronize=True)
    model.set_profiling(1)
    for _ in range(2): model.calc_polarizabilities_device(pos, synchronize=True)
    t = model.kernel_times(); tot = sum(v[0] for v in t.values())
    print(name, {k: round(v[0]/2000*1e3,2) for k,v in t.items() if v[0]>0}, "total us/structure", round(tot/2000*1e3,2))
