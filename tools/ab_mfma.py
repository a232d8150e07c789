"""A/B in ONE process: the fused pipeline with split-f16 MFMA vs exact-f32 MFMA, interleaved
rounds, per-kernel HIP-event times.  usage: python tools/ab_mfma.py [cells] [frames] [rounds]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload  # noqa: E402


def main():
    cells = tuple(int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "4,4,2").split(","))
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    wl = make_workload(cells, frames, "perf", seed=33)
    models = {}
    for name, env in (("f16", None), ("f32", "f32")):
        if env:
            os.environ["RN_POTGNN_MFMA"] = env
        else:
            os.environ.pop("RN_POTGNN_MFMA", None)
        models[name] = wl["model"]()
        models[name]._ensure_handle()
    pos = torch.tensor(wl["positions"], device="cuda")
    out = {k: torch.empty((frames, 3, 3), dtype=torch.float64, device="cuda") for k in models}
    for k, m in models.items():
        m.calc_polarizabilities_device(pos, out[k])
    torch.cuda.synchronize()
    d = (out["f16"] - out["f32"]).abs().max().item() / out["f32"].abs().max().item()
    print(f"max rel diff f16 vs f32: {d:.2e}")
    times = {k: [] for k in models}
    for _ in range(rounds):
        for k, m in models.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.calc_polarizabilities_device(pos, out[k])
            torch.cuda.synchronize()
            times[k].append(time.perf_counter() - t0)
    for k in models:
        t = np.array(times[k])
        print(f"{k}: median {np.median(t)*1e3:.2f} ms  min {t.min()*1e3:.2f} ms  -> {frames/np.median(t):.0f} structures/s")
    for k, m in models.items():
        m.set_profiling(1)
        m.calc_polarizabilities_device(pos, out[k], synchronize=True)
        kt = m.kernel_times()
        print(k, {n: round(v[0] / frames * 1e3, 3) for n, v in kt.items() if v[1]}, "us per structure")
        m.set_profiling(0)


if __name__ == "__main__":
    main()
