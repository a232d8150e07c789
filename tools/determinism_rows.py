"""Where do irreproducible EdgeBlock rows sit?  (race localisation, second step)

Evaluates the same frames twice with stage snapshots and, for the first EdgeBlock output that
differs, lists every differing row with its position in the fused kernel's schedule: tile, index in
the tile's destination list, round, lane group (slot), and which columns moved by how much.
usage: RN_POTGNN_LIB=... python3 tools/determinism_rows.py [frames] [reps]
"""
import os
import sys
from collections import Counter

import numpy as np

os.environ["RN_POTGNN_KEEP_STAGES"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = make_workload((4, 4, 2), frames, "perf", seed=33)
model = wl["model"](max_chunk_structures=frames)
ei = model.ref_edge_indexes
ea, eb = ei[1], ei[2]
E = ea.size
N = model.num_atoms
atoms_per_tile = 4
# destination lists per tile: in-edges of the tile's atoms, atom by atom, ascending edge id
order = np.lexsort((np.arange(E), eb))
pos_in_tile = np.empty(E, dtype=np.int64)
tile_of = eb // atoms_per_tile
for t in range(N // atoms_per_tile):
    d = order[(eb[order] // atoms_per_tile) == t]
    pos_in_tile[d] = np.arange(d.size)


def stage(p):
    model.calc_polarizabilities(wl["positions"])
    return model.debug_stage(2, p)


for p in (1,):
    ref = stage(p)
    for rep in range(reps):
        cur = stage(p)
        d = np.abs(cur - ref)
        bad = np.argwhere(d.max(axis=1) > 0)[:, 0]
        print(f"pass {p} rep {rep}: {bad.size} rows differ in {np.unique(bad // E).size} frames", flush=True)
        slots, rounds, mods, tiles = Counter(), Counter(), Counter(), Counter()
        for row in bad[:4000]:
            f, e = divmod(int(row), E)
            i = int(pos_in_tile[e])
            big = np.nonzero(d[row] > 0.25 * d[row].max())[0]
            slots[i % 16] += 1
            rounds[i // 16] += 1
            tiles[int(tile_of[e])] += 1
            mods[tuple(sorted(set((big % 4).tolist())))] += 1
        print("  by lane group (dest slot):", sorted(slots.items()))
        print("  by round:", sorted(rounds.items()))
        print("  big-deviation columns mod 4:", mods.most_common(6))
        print("  tiles hit:", len(tiles), "most common", tiles.most_common(5))
        fr = Counter((bad // E).tolist())
        print("  rows per frame (top):", fr.most_common(5), " frames mod 8:", sorted(Counter((np.unique(bad // E) % 8).tolist()).items()))
        for row in bad[:5]:
            f, e = divmod(int(row), E)
            big = np.nonzero(d[row] > 0.25 * d[row].max())[0]
            print(f"    frame {f} edge {e} (a={ea[e]} b={eb[e]}) tile {tile_of[e]} i={pos_in_tile[e]} max {d[row].max():.3g} big cols {big.tolist()[:20]}")
        # do differing rows share a source row?  (a wrong Q' row would hit every destination of its atom)
        per_atom = Counter(zip((bad // E).tolist(), eb[bad % E].tolist()))
        print("  rows per (frame, destination atom):", Counter(per_atom.values()).most_common(6), flush=True)
