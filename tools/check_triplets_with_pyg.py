#!/usr/bin/env python3
"""Pin the ONE point of the oracle that this repository could not check against the real thing: the ORDER in which
``torch_geometric.nn.models.dimenet.triplets`` lists edge triplets (ramannoodle/pmodel/torch/_utils.py:157-160 calls it with
``edge_index = ref_edge_indexes[[1, 2]]``; ``_gnn.py:647-650`` consumes the 7-tuple positionally).  The build container has
no torch_geometric, so ``tests/golden/_standins.py`` and ``oracle/potgnn_oracle.py: triplets`` restate its published
algorithm, and every fixture's ``trip/*`` arrays come from that restatement.

Run this on any machine that HAS torch_geometric (>= 2.3, as the reference's pyproject pins):

    python tools/check_triplets_with_pyg.py

For every fixture under tests/golden/ that carries ``ref_edge_indexes`` it calls the real ``triplets`` and compares all seven
arrays with the fixture's, then compares the fixtures' SHA-256 digests with tests/golden/triplet_hashes.json (which
``tests/test_oracle_golden.py::test_triplet_fixture_digests`` also checks, without PyG).  Exit status 0 = the restated
ordering IS PyG's; 1 = it is not (the set of triplets is fixed by the graph, so a mismatch changes only the fp32 summation
order of the EdgeBlock's scatter -- but then "bit-exact triplet indices" in DESIGN.md section 2 has to be re-stated)."""
import glob
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("i", "j", "idx_i", "idx_j", "idx_k", "slot5", "slot6")  # the 7-tuple in PyG's order (col, row, idx_i, idx_j, idx_k, idx_kj, idx_ji)


def digest(g):
    """SHA-256 over the seven arrays as little-endian int64, in tuple order."""
    h = hashlib.sha256()
    for k in KEYS:
        h.update(np.ascontiguousarray(g["trip/" + k], dtype="<i8").tobytes())
    return h.hexdigest()


def fixtures():
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
        g = np.load(path)
        if "ref_edge_indexes" in g.files and all("trip/" + k in g.files for k in KEYS):
            yield os.path.basename(path), g


def main():
    recorded = json.load(open(os.path.join(ROOT, "tests", "golden", "triplet_hashes.json")))
    try:
        import torch
        from torch_geometric.nn.models.dimenet import triplets
    except ImportError as exc:
        print(f"torch_geometric is not importable here ({exc}): nothing was checked against PyG.")
        print("digests of the fixtures vs tests/golden/triplet_hashes.json:")
        bad = [n for n, g in fixtures() if recorded.get(n) != digest(g)]
        print("  all match" if not bad else f"  MISMATCH: {bad}")
        return 2
    import torch_geometric
    print(f"torch_geometric {torch_geometric.__version__}")
    failures = 0
    for name, g in fixtures():
        edges = torch.as_tensor(g["ref_edge_indexes"][[1, 2]], dtype=torch.long)
        num_nodes = int(g["positions"].shape[0])
        real = triplets(edges, num_nodes=num_nodes)
        ok = len(real) == 7
        for k, arr in zip(KEYS, real):
            same = np.array_equal(np.asarray(arr.cpu(), dtype=np.int64), np.asarray(g["trip/" + k], dtype=np.int64))
            ok = ok and same
            if not same:
                print(f"  {name}: trip/{k} differs from torch_geometric's")
        ok = ok and recorded.get(name) == digest(g)
        print(f"{name:28s} E={edges.shape[1]:6d} T={len(g['trip/idx_i']):8d}  {'identical to PyG' if ok else 'DIFFERENT'}")
        failures += not ok
    print("the restated triplet ordering IS torch_geometric's" if not failures else f"{failures} fixture(s) differ")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
