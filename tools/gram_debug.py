"""Debug aid: which EdgeBlock rows differ between the GRAM and the plain instantiation of the role-specialised kernel."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
os.environ["RN_POTGNN_KEEP_STAGES"] = "1"
from tests.conftest import load_golden
from tests.test_gpu_parity import _random_model

case, cutoff = (sys.argv[1], float(sys.argv[2])) if len(sys.argv) > 2 else ("rocksalt64_parity", 3.2)
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 3
g = load_golden(case)
pos = g["pos_batch"][:frames]
out = {}
for gram in ("0", "1"):
    os.environ["RN_POTGNN_PS_GRAM"] = gram
    model, oracle = _random_model(g, cutoff, 64, 64, 1, seed=5)
    model.eval()
    s = pos.shape[0]
    lat = torch.tensor(g["lattice"]).expand(s, 3, 3)
    zs = torch.tensor(g["atomic_numbers"]).expand(s, -1)
    model.forward(lat, zs, torch.tensor(pos))
    out[gram] = model.debug_stage(2, 1)
    print("gram", gram, model.config_flags()["role_split_edge_block"], out[gram].shape)
e = model.num_edges
edges = model.ref_edge_indexes
d = np.abs(out["1"] - out["0"]).max(axis=1)
bad = np.nonzero(d > 1e-5)[0]
print("rows differing:", len(bad), "of", len(d), "max", d.max())
for r in bad[:40]:
    f, ed = divmod(r, e)
    print("frame", f, "edge", ed, "a", edges[1][ed], "b", edges[2][ed], "err", d[r], "cols", np.nonzero(np.abs(out["1"][r] - out["0"][r]) > 1e-5)[0][:8])
# per-b-atom summary
if len(bad):
    bs = edges[2][bad % e]
    print("b atoms of bad rows:", np.unique(bs)[:50])
    print("frames of bad rows:", np.unique(bad // e))
