import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, bench
for hp, frames in (("parity", 60000), ("perf", 25000)):
    wl = bench.make_workload(num_cells=(4, 4, 2), frames=frames, hparams=hp, seed=5)
    model = wl["model"](device=0)
    pos = wl["positions"]
    t = time.perf_counter(); a = model.calc_polarizabilities(pos); dt = time.perf_counter() - t
    assert a.shape == (frames, 3, 3) and np.isfinite(a).all()
    idx = np.random.default_rng(0).choice(frames, 500, replace=False)
    b = model.calc_polarizabilities(pos[idx])
    assert np.array_equal(a[idx], b), "subset differs"
    e = model.calc_polarizabilities(pos[:0]); assert e.shape == (0, 3, 3)
    one = model.calc_polarizabilities(pos[7:8]); assert np.array_equal(one[0], a[7])
    print(hp, frames, "frames ok,", round(frames / dt), "structures/s from the host array; peak HBM", torch.cuda.max_memory_allocated() >> 20, "MiB (torch only)")
