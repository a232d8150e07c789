"""Round-3 golden fixtures, produced by running the REFERENCE implementation (read-only import).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r3.py

* ``triclinic20_r3.npz`` -- the triclinic20 model of round 1 (same seeds, so the same weights; the
  script asserts it) through ``PotGNN.forward`` with ``atomic_numbers`` that DIFFER BETWEEN SAMPLES
  (``_convert_to_atom_type`` + ``_node_embedding``, ``_gnn.py:541-557,642-643``): sample 0 carries the
  reference structure's species, the others have pairs of atoms of different species swapped, one
  has every atom of one species replaced by another, and half of them also carry a strained lattice.
  float32 and float64 outputs.
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import _standins  # noqa: E402

_standins.install()

import make_golden as R1  # noqa: E402  (imports the reference; its main() is not run)
from make_golden_r2 import to_f64  # noqa: E402

torch.set_num_threads(4)


def main():
    old = np.load(os.path.join(HERE, "triclinic20.npz"))
    rng = np.random.default_rng(404)
    lattice, positions, zs = R1.triclinic(rng)
    hp = dict(cutoff=3.0, fn=8, fe=12, passes=2, g0=0.0, g1=4.0)
    rng = np.random.default_rng(404)
    sym = rng.normal(size=(3, 3))
    mean = (sym + sym.T) * 2.0 + np.diag([40.0, 41.0, 39.0])
    std = np.abs(rng.normal(size=(3, 3)))
    std = (std + std.T) * 0.5 + 0.2
    _, model = R1.build(lattice, positions, zs, hp, mean, std, 404, "soft", torch.float32)
    for k, v in model.state_dict().items():
        assert np.array_equal(v.numpy(), old["sd/" + k]), k
    model64 = to_f64(model, lattice, positions, zs, hp, mean, std, 404, "soft")

    rng = np.random.default_rng(5151)
    s = 7
    pos = old["pos_batch"][rng.integers(0, len(old["pos_batch"]), s)]
    zz = np.tile(np.array(zs, dtype=np.int32), (s, 1))
    for i in range(1, s - 1):  # swap pairs of atoms of different species
        for _ in range(i):
            a, b = rng.choice(len(zs), 2, replace=False)
            while zz[i, a] == zz[i, b]:
                a, b = rng.choice(len(zs), 2, replace=False)
            zz[i, a], zz[i, b] = zz[i, b], zz[i, a]
    zz[s - 1][zz[s - 1] == 8] = 38  # a whole species replaced
    lats = np.tile(lattice, (s, 1, 1))
    strain = np.eye(3)[None] + rng.uniform(-0.03, 0.03, (s, 3, 3))
    lats[1::2] = np.einsum("ij,sjk->sik", lattice, strain[1::2])
    zt = torch.tensor(zz, dtype=torch.int)
    model.eval()
    with torch.no_grad():
        out = model.forward(torch.tensor(lats, dtype=torch.float32), zt, torch.tensor(pos, dtype=torch.float32))
        same = model.forward(torch.tensor(lats, dtype=torch.float32),
                             torch.tensor(zs, dtype=torch.int).expand(s, -1), torch.tensor(pos, dtype=torch.float32))
    assert np.array_equal(out.numpy()[0], same.numpy()[0])
    assert np.abs(out.numpy()[1:] - same.numpy()[1:]).max() > 1e-3  # the species matter
    torch.set_default_dtype(torch.float64)
    model64.eval()
    with torch.no_grad():
        out64 = model64.forward(torch.tensor(lats), zt, torch.tensor(pos))
    torch.set_default_dtype(torch.float32)
    print("triclinic20_r3: forward with per-sample species, f32 vs f64:",
          float(np.abs(out.numpy() - out64.numpy()).max() / np.abs(out64.numpy()).max()))
    np.savez_compressed(os.path.join(HERE, "triclinic20_r3.npz"), **{
        "zs/atomic_numbers": zz, "zs/lattices": lats, "zs/positions": pos,
        "zs/forward": out.numpy(), "zs/forward64": out64.numpy()})


if __name__ == "__main__":
    main()
