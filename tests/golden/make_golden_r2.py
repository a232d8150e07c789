"""Round-2 golden fixtures, produced by running the REFERENCE implementation (read-only import),
next to the round-1 set of ``make_golden.py`` (which stays bit-for-bit as committed).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r2.py

* ``triclinic20_r2.npz`` -- the triclinic20 model of round 1 (same seeds, so the same weights;
  the script asserts it) with
    - ``lat/*``: ``PotGNN.forward`` with a DIFFERENT lattice per sample (``_gnn.py:603-611``),
    - ``md64/*``: the MD trajectory of the round-1 fixture through the float64 reference model
      and ``MDRamanSpectrum.measure`` (what the float32 spectrum must be judged against),
    - ``train64/*``: one training step of the float64 reference model (outputs, loss, gradients).
* ``config1_plumbing.npz`` -- BASELINE config 1: 8-atom rocksalt cell, a seeded *linear*
  ``PolarizabilityModel`` (what an order-1 P1 ``InterpolationModel`` computes,
  ``pmodel/_interpolation.py:239-244``), 24 modes through the reference's ``Phonons`` ->
  ``PhononRamanSpectrum.measure``.
* ``perf256_frames.npz`` -- two frames of the 256-atom perf configuration (BASELINE config 3:
  rocksalt 4x4x2, Fn = Fe = 64, P = 4) through the reference's ``calc_polarizabilities`` in
  float32 and float64; inputs are regenerated from ``bench.make_workload`` by the test.
"""
from __future__ import annotations

import copy
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
from ramannoodle.abstract import PolarizabilityModel  # noqa: E402
from ramannoodle.dynamics._phonon import Phonons  # noqa: E402
from ramannoodle.dynamics._trajectory import Trajectory  # noqa: E402

torch.set_num_threads(4)


def to_f64(model, lattice, positions, zs, hp, mean, std, seed, style):
    _, model64 = R1.build(lattice, positions, zs, hp, mean, std, seed, style, torch.float64)
    sd = {k: v.double() if v.is_floating_point() else v for k, v in model.state_dict().items()}
    torch.set_default_dtype(torch.float64)
    model64.load_state_dict(sd)
    torch.set_default_dtype(torch.float32)
    return model64


def triclinic_r2():
    old = np.load(os.path.join(HERE, "triclinic20.npz"))
    rng = np.random.default_rng(404)
    lattice, positions, zs = R1.triclinic(rng)
    hp = dict(cutoff=3.0, fn=8, fe=12, passes=2, g0=0.0, g1=4.0)
    # same construction as make_case("triclinic20", seed=404)
    rng = np.random.default_rng(404)
    sym = rng.normal(size=(3, 3))
    mean = (sym + sym.T) * 2.0 + np.diag([40.0, 41.0, 39.0])
    std = np.abs(rng.normal(size=(3, 3)))
    std = (std + std.T) * 0.5 + 0.2
    ref, model = R1.build(lattice, positions, zs, hp, mean, std, 404, "soft", torch.float32)
    for k, v in model.state_dict().items():
        assert np.array_equal(v.numpy(), old["sd/" + k]), k
    model64 = to_f64(model, lattice, positions, zs, hp, mean, std, 404, "soft")
    data = {}
    rng = np.random.default_rng(4242)

    # ---- per-sample lattices (strained by up to +-3 %, sheared)
    s = 6
    pos = old["pos_batch"][rng.integers(0, len(old["pos_batch"]), s)]
    strain = np.eye(3)[None] + rng.uniform(-0.03, 0.03, (s, 3, 3))
    lats = np.einsum("ij,sjk->sik", lattice, strain)
    lats[0] = lattice
    zz = torch.tensor(zs, dtype=torch.int).unsqueeze(0).expand(s, -1)
    model.eval()
    with torch.no_grad():
        out = model.forward(torch.tensor(lats, dtype=torch.float32), zz, torch.tensor(pos, dtype=torch.float32))
        same = model.forward(torch.tensor(lattice, dtype=torch.float32).expand(s, 3, 3), zz,
                             torch.tensor(pos, dtype=torch.float32))
    assert np.abs(out.numpy()[1:] - same.numpy()[1:]).max() > 1e-3  # the lattice matters
    data.update({"lat/lattices": lats, "lat/positions": pos, "lat/forward": out.numpy()})
    torch.set_default_dtype(torch.float64)
    model64.eval()
    with torch.no_grad():
        out64 = model64.forward(torch.tensor(lats), zz, torch.tensor(pos))
    torch.set_default_dtype(torch.float32)
    data["lat/forward64"] = out64.numpy()

    # ---- MD spectrum through the float64 model (same trajectory as md/* of round 1)
    torch.set_default_dtype(torch.float64)
    md = Trajectory(old["md/positions"], float(old["md/timestep"])).get_raman_spectrum(model64)
    torch.set_default_dtype(torch.float32)
    data["md64/alpha_ts"] = md.polarizability_ts
    w, i0 = md.measure()
    data["md64/wavenumbers"] = w
    data["md64/int_raw"] = i0
    rel = np.abs(i0 - old["md/int_raw"]).max() / np.abs(i0).max()
    print(f"triclinic20: reference f32 vs f64 MD spectrum, max diff / scale = {rel:.2e}")
    data["md64/ref_f32_vs_f64"] = np.float64(rel)

    # ---- one training step in float64 (same batch and targets as triclinic20_train)
    tr = np.load(os.path.join(HERE, "triclinic20_train.npz"))
    m = copy.deepcopy(model64)
    m.train()
    s = tr["train/target"].shape[0]
    torch.set_default_dtype(torch.float64)
    lat = torch.tensor(lattice).unsqueeze(0).expand(s, -1, -1)
    zz = torch.tensor(zs, dtype=torch.int).unsqueeze(0).expand(s, -1)
    out = m.forward(lat, zz, torch.tensor(tr["pos_batch"][:s]))
    loss = torch.nn.MSELoss()(out, torch.tensor(tr["train/target"]).double())
    loss.backward()
    torch.set_default_dtype(torch.float32)
    data["train64/out"] = out.detach().numpy()
    data["train64/loss"] = np.float64(loss.item())
    worst = 0.0
    for k, p in m.named_parameters():
        g = p.grad.detach().numpy().copy()
        data["train64/grad/" + k] = g
        worst = max(worst, np.abs(g - tr["train/grad/" + k]).max() / max(np.abs(g).max(), 1e-30))
    print(f"triclinic20_train: reference f32 vs f64 gradients, worst max diff / max = {worst:.2e}")
    data["train64/ref_f32_vs_f64"] = np.float64(worst)
    np.savez_compressed(os.path.join(HERE, "triclinic20_r2.npz"), **data)


def triclinic_randn():
    """The reference's own batch test draws positions from N(0,1) -- far outside the unit cell
    (``test/tests/torch/test_gnn.py:83-113``): ``forward`` of the triclinic20 model on such positions,
    batch sizes 1..3 in one array, float32 and float64."""
    old = np.load(os.path.join(HERE, "triclinic20.npz"))
    rng = np.random.default_rng(404)
    lattice, positions, zs = R1.triclinic(rng)
    hp = dict(cutoff=3.0, fn=8, fe=12, passes=2, g0=0.0, g1=4.0)
    rng = np.random.default_rng(404)
    sym = rng.normal(size=(3, 3))
    mean = (sym + sym.T) * 2.0 + np.diag([40.0, 41.0, 39.0])
    std = np.abs(rng.normal(size=(3, 3)))
    std = (std + std.T) * 0.5 + 0.2
    ref, model = R1.build(lattice, positions, zs, hp, mean, std, 404, "soft", torch.float32)
    for k, v in model.state_dict().items():
        assert np.array_equal(v.numpy(), old["sd/" + k]), k
    model64 = to_f64(model, lattice, positions, zs, hp, mean, std, 404, "soft")
    torch.manual_seed(2024)
    pos = torch.randn(6, len(zs), 3)
    lat = torch.tensor(lattice, dtype=torch.float32).expand(6, 3, 3)
    zz = torch.tensor(zs, dtype=torch.int).unsqueeze(0).expand(6, -1)
    model.eval()
    with torch.no_grad():
        whole = model.forward(lat, zz, pos)
        single = torch.cat([model.forward(lat[i:i + 1], zz[i:i + 1], pos[i:i + 1]) for i in range(6)])
    assert torch.allclose(whole, single, atol=1e-5)  # the reference's own assertion
    torch.set_default_dtype(torch.float64)
    model64.eval()
    with torch.no_grad():
        whole64 = model64.forward(lat.double(), zz, pos.double())
    torch.set_default_dtype(torch.float32)
    assert np.isfinite(whole.numpy()).all()
    np.savez_compressed(os.path.join(HERE, "triclinic20_randn.npz"), positions=pos.numpy(),
                        forward=whole.numpy(), forward64=whole64.numpy(),
                        calc=model.calc_polarizabilities(pos.numpy().astype(np.float64)))
    print(f"triclinic20_randn: |x| up to {pos.abs().max():.2f}, f32 vs f64 "
          f"{np.abs(whole.numpy() - whole64.numpy()).max():.2e}")


def tio2_gnn_test():
    """The configuration of the reference's own PotGNN tests (``test/tests/torch/test_gnn.py:170-193``:
    ``PotGNN(ref_structure, 2, 5, 5, 5, 0, 5, ...)`` on test/data/TiO2/POSCAR -- cutoff 2 A,
    Fn = Fe = 5, five message passes), captured like the round-1 cases (same file layout)."""
    import ramannoodle as rn
    ref = rn.io.vasp.poscar.read_ref_structure("/root/reference/test/data/TiO2/POSCAR")
    hp = dict(cutoff=2.0, fn=5, fe=5, passes=5, g0=0.0, g1=5.0)
    stages = ["unit", "dist", "node0", "edge0"] + [f"{k}{p}" for p in range(1, 6) for k in ("node", "edge")] + ["pol_emb"]
    R1.make_case("tio2_gnn_test", ref.lattice, ref.positions, ref.atomic_numbers, hp, seed=606, style="notebook",
                 s=3, keep=stages)


class LinearModel(PolarizabilityModel):
    """alpha(x) = alpha0 + sum_k c_k (x - x0)_k  -- a seeded linear map, symmetric tensors."""

    def __init__(self, ref_positions, alpha0, coeff):
        self.x0, self.alpha0, self.coeff = ref_positions, alpha0, coeff

    def calc_polarizabilities(self, positions_batch):
        d = (positions_batch - self.x0[None]).reshape(positions_batch.shape[0], -1)
        return self.alpha0[None] + np.einsum("sk,kij->sij", d, self.coeff)


def config1():
    lattice, positions, zs = R1.rocksalt(1, 1, 1)
    rng = np.random.default_rng(11)
    n = len(zs)
    coeff = rng.normal(size=(3 * n, 3, 3))
    coeff = coeff + np.swapaxes(coeff, 1, 2)
    alpha0 = np.diag([5.0, 5.5, 6.0])
    qmat, _ = np.linalg.qr(rng.normal(size=(3 * n, 3 * n)))
    mass = np.where(np.array(zs) == 12, 24.305, 15.999)
    disp = (qmat.T.reshape(3 * n, n, 3) / np.sqrt(mass)[None, :, None]) @ np.linalg.inv(lattice)
    wn = np.linspace(50.0, 900.0, 3 * n)
    spec = Phonons(positions, wn, disp).get_raman_spectrum(LinearModel(positions, alpha0, coeff))
    data = dict(lattice=lattice, positions=positions, atomic_numbers=np.array(zs, dtype=np.int32),
                coeff=coeff, alpha0=alpha0, displacements=disp, wavenumbers=wn,
                raman_tensors=spec.raman_tensors)
    w, i0 = spec.measure()
    data["out_wavenumbers"], data["int_raw"] = w, i0
    w, i1 = spec.measure(laser_correction=True, laser_wavelength=522, bose_einstein_correction=True,
                         temperature=300)
    data["int_corr"] = i1
    np.savez_compressed(os.path.join(HERE, "config1_plumbing.npz"), **data)
    print(f"config1: N={n}, M={len(wn)}, max intensity {i0.max():.3e}")


def perf256():
    # the workload definition (cell, frames, weight initialisation) is shared with bench.py
    from bench import HPARAMS, md_frames, rocksalt, synthetic_state
    from ramannoodle.pmodel.torch import PotGNN
    from ramannoodle.structure._reference import ReferenceStructure
    fn, fe, passes = HPARAMS["perf"]
    lattice, ref, zs = rocksalt(4, 4, 2)
    rng = np.random.default_rng(33)
    positions = md_frames(rng, lattice, ref, 2)
    sym = rng.normal(size=(3, 3))
    mean = (sym + sym.T) + np.diag([30.0, 31.0, 29.0])
    std = np.abs(rng.normal(size=(3, 3)))
    std = (std + std.T) * 0.5 + 0.2
    torch.manual_seed(7)
    model = PotGNN(ReferenceStructure(list(zs), lattice, ref), 3.2, fn, fe, passes, 0.0, 5.0, mean, std)
    state = synthetic_state(model, 7)
    model.load_state_dict(state)
    alpha32 = model.calc_polarizabilities(positions)
    model64 = to_f64(model, lattice, ref, zs, dict(cutoff=3.2, fn=fn, fe=fe, passes=passes, g0=0.0, g1=5.0),
                     mean, std, 7, "notebook")
    torch.set_default_dtype(torch.float64)
    alpha64 = model64.calc_polarizabilities(positions)
    torch.set_default_dtype(torch.float32)
    print(f"perf256: E={model._ref_edge_indexes.shape[1]}, f32 vs f64 "
          f"{np.abs(alpha32 - alpha64).max() / np.abs(alpha64).max():.2e}")
    np.savez_compressed(os.path.join(HERE, "perf256_frames.npz"), positions=positions, mean=mean, std=std,
                        alpha32=alpha32, alpha64=alpha64,
                        probe_weight=state["_edge_blocks.3.c3_linear.weight"][:2, :8].numpy())


if __name__ == "__main__":
    if sys.argv[1:] == ["randn"]:  # (added after the others: leaves their files untouched)
        triclinic_randn()
    elif sys.argv[1:] == ["tio2_gnn_test"]:
        tio2_gnn_test()
    else:
        triclinic_r2()
        config1()
        perf256()
        triclinic_randn()
        tio2_gnn_test()
