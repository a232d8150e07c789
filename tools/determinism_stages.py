"""Which stage first differs between two evaluations of the same frames? (race localisation)"""
import os
import sys

import numpy as np

os.environ["RN_POTGNN_KEEP_STAGES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wl = make_workload((4, 4, 2), frames, "perf", seed=33)
model = wl["model"](max_chunk_structures=frames)


def stages():
    model.calc_polarizabilities(wl["positions"])
    out = {}
    for p in range(5):
        out[f"node{p}"] = model.debug_stage(1, p)
        out[f"edge{p}"] = model.debug_stage(2, p)
    out["pol"] = model.debug_stage(3)
    return out


ref = stages()
order = ["node0", "edge0"] + [f"{k}{p}" for p in range(1, 5) for k in ("node", "edge")] + ["pol"]
for r in range(reps):
    cur = stages()
    line = []
    for k in order:
        d = np.abs(cur[k] - ref[k])
        nbad = int((d.reshape(frames, -1).max(axis=1) > 0).sum())
        line.append(f"{k}:{nbad}")
        if nbad and "--detail" in sys.argv:
            rows = np.argwhere(d > 0)
            print("   ", k, "first differing (row, col):", rows[:6].tolist(), "max", d.max())
            if k.startswith("edge") and "--dump" in sys.argv:
                r0 = rows[0][0]
                np.set_printoptions(linewidth=250, precision=4, suppress=True)
                print("      cols differing:", sorted(set(rows[rows[:, 0] == r0][:, 1].tolist())))
                print("      ref :", ref[k][r0])
                print("      cur :", cur[k][r0])
                # does the wrong row equal some other row of the reference (same frame)?
                f = r0 // 4608
                blk = ref[k][f * 4608:(f + 1) * 4608]
                dist = np.abs(blk - cur[k][r0][None]).max(axis=1)
                print("      nearest reference row in the frame:", int(dist.argmin()), "dist", float(dist.min()), "own edge", r0 % 4608)
                sys.argv.remove("--dump")
    print(" ".join(line), flush=True)
