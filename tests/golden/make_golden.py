"""Generate golden fixtures by running the REFERENCE implementation (read-only import).

Run in the build container only (``/root/reference`` mounted)::

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's own code -- ``ramannoodle/pmodel/torch/_gnn.py`` (PotGNN),
``_utils.py`` (graph/triplets batching), ``dynamics/_phonon.py``,
``dynamics/_trajectory.py``, ``spectrum/_raman.py`` -- is imported unmodified; only the
absent third-party symbols are stood in (see ``_standins.py``).  Outputs are small
``.npz`` files of inputs + expected outputs committed next to this script.  Nothing from
the reference is copied: fixtures are data.
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

import _standins  # noqa: E402

_standins.install()

import ramannoodle as rn  # noqa: E402
from ramannoodle.pmodel.torch import PotGNN  # noqa: E402
from ramannoodle.dataset.torch.utils import polarizability_vectors_to_tensors  # noqa: E402
from ramannoodle.structure._reference import ReferenceStructure  # noqa: E402
from ramannoodle.dynamics._phonon import Phonons  # noqa: E402
from ramannoodle.dynamics._trajectory import Trajectory  # noqa: E402

torch.set_num_threads(4)


# --------------------------------------------------------------------------- cells
def rocksalt(nx, ny, nz, a=4.2):
    """Rocksalt supercell; Z=12 on even-parity sites, Z=8 on odd."""
    half = a / 2.0
    pos, zs = [], []
    for ix in range(2 * nx):
        for iy in range(2 * ny):
            for iz in range(2 * nz):
                pos.append([ix / (2.0 * nx), iy / (2.0 * ny), iz / (2.0 * nz)])
                zs.append(12 if (ix + iy + iz) % 2 == 0 else 8)
    lattice = np.diag([2 * nx * half, 2 * ny * half, 2 * nz * half]).astype(np.float64)
    return lattice, np.array(pos, dtype=np.float64), zs


def triclinic(rng, num_atoms=20):
    lattice = np.array([[6.3, 0.4, -0.3], [0.9, 6.8, 0.5], [-0.6, 0.7, 7.1]])
    # jittered grid so that no two atoms overlap
    grid = np.array(
        [[i, j, k] for i in range(3) for j in range(3) for k in range(3)], dtype=float
    )
    sel = rng.permutation(27)[:num_atoms]
    pos = (grid[sel] + 0.5 + rng.uniform(-0.22, 0.22, (num_atoms, 3))) / 3.0
    zs = [int(z) for z in rng.choice([22, 8, 38], num_atoms)]
    zs[0], zs[1], zs[2] = 22, 8, 38
    return lattice, pos, zs


# --------------------------------------------------------------------------- model
def randomize(model, seed, style):
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            is_norm = "norm" in name or "_to_polarizability_embedding.1." in name
            if is_norm and name.endswith("weight"):
                p.copy_(torch.rand(p.shape, generator=gen) + 0.5)
            elif is_norm and name.endswith("bias"):
                p.copy_(torch.rand(p.shape, generator=gen) * 0.6 - 0.3)
            elif name.endswith("bias"):
                p.copy_(torch.rand(p.shape, generator=gen) - 0.5)
            elif style == "notebook" or p.dim() < 2 or "_node_embedding.0" in name:
                p.copy_(torch.randn(p.shape, generator=gen))
            else:
                p.copy_(torch.randn(p.shape, generator=gen) / np.sqrt(p.shape[1]))
        bn = model._to_polarizability_embedding[1]
        bn.running_mean.copy_(torch.randn(bn.running_mean.shape, generator=gen) * 0.3)
        bn.running_var.copy_(torch.rand(bn.running_var.shape, generator=gen) + 0.5)


def build(lattice, positions, zs, hp, mean, std, seed, style, dtype):
    torch.set_default_dtype(dtype)
    ref = ReferenceStructure(list(zs), lattice, positions)
    model = PotGNN(
        ref, hp["cutoff"], hp["fn"], hp["fe"], hp["passes"], hp["g0"], hp["g1"], mean, std
    )
    torch.set_default_dtype(torch.float32)
    if dtype == torch.float32:
        randomize(model, seed, style)
    return ref, model


def capture(model, lattice, zs, pos_batch, dtype):
    """Run reference forward with hooks; return dict of intermediates."""
    torch.set_default_dtype(dtype)
    out = {}
    hooks = []

    def save(key):
        def fn(_m, _i, o):
            out[key] = o.detach().numpy().copy()

        return fn

    hooks.append(model._node_embedding.register_forward_hook(save("node0")))
    hooks.append(model._edge_embedding.register_forward_hook(save("edge0")))
    for p, (nb, eb) in enumerate(zip(model._node_blocks, model._edge_blocks)):
        hooks.append(nb.register_forward_hook(save(f"node{p + 1}")))
        hooks.append(eb.register_forward_hook(save(f"edge{p + 1}")))
    hooks.append(model._to_polarizability_embedding.register_forward_hook(save("pol_emb")))
    model.eval()
    s = pos_batch.shape[0]
    lat = torch.tensor(lattice).unsqueeze(0).expand(s, -1, -1).type(dtype)
    zz = torch.tensor(zs, dtype=torch.int).unsqueeze(0).expand(s, -1)
    with torch.no_grad():
        edge_index, unit, dist = model._batch_graph(lat, torch.tensor(pos_batch).type(dtype))
        out["unit"] = unit.numpy().copy()
        out["dist"] = dist.numpy().copy()
        fwd = model.forward(lat, zz, torch.tensor(pos_batch).type(dtype))
    out["forward"] = fwd.numpy().copy()
    for h in hooks:
        h.remove()
    torch.set_default_dtype(torch.float32)
    return out


def model_arrays(model):
    d = {}
    for k, v in model.state_dict().items():
        d["sd/" + k] = v.detach().numpy().copy()
    d["ref_edge_indexes"] = model._ref_edge_indexes.numpy().copy()
    for n, t in zip(
        ["i", "j", "idx_i", "idx_j", "idx_k", "slot5", "slot6"], model._batch_triplets._ref_triplets
    ):
        d["trip/" + n] = t.numpy().copy()
    d["atom_type_map"] = model._atom_type_map.numpy().copy()
    d["gauss_coefficient"] = np.float64(model._edge_embedding.coefficient)
    return d


def frames(rng, positions, lattice, s, amp=0.05):
    cart = rng.normal(0.0, amp, (s,) + positions.shape)
    frac = cart @ np.linalg.inv(lattice)
    x = positions[None] + frac
    return x - np.floor(x)


def make_case(name, lattice, positions, zs, hp, seed, style, s, keep, with_f64=True,
              extras=None):
    rng = np.random.default_rng(seed)
    sym = rng.normal(size=(3, 3))
    mean = (sym + sym.T) * 2.0 + np.diag([40.0, 41.0, 39.0])
    std = np.abs(rng.normal(size=(3, 3)))
    std = (std + std.T) * 0.5 + 0.2
    ref, model = build(lattice, positions, zs, hp, mean, std, seed, style, torch.float32)
    pos_batch = frames(rng, positions, lattice, s)
    cap = capture(model, lattice, zs, pos_batch, torch.float32)
    data = model_arrays(model)
    data.update(
        lattice=lattice, positions=positions, atomic_numbers=np.array(zs, dtype=np.int32),
        mean=mean, std=std, pos_batch=pos_batch,
        hp=np.array([hp["cutoff"], hp["fn"], hp["fe"], hp["passes"], hp["g0"], hp["g1"]]),
    )
    for k in keep:
        data["f32/" + k] = cap[k]
    data["f32/forward"] = cap["forward"]
    data["f32/alpha"] = model.calc_polarizabilities(pos_batch)
    e = model._ref_edge_indexes.shape[1]
    t = model._batch_triplets._num_triplets
    print(f"{name}: N={len(zs)} E={e} T={t} params={sum(p.numel() for p in model.parameters())}")

    if with_f64:
        _, model64 = build(lattice, positions, zs, hp, mean, std, seed, style, torch.float64)
        sd = {k: v.double() if v.is_floating_point() else v for k, v in model.state_dict().items()}
        torch.set_default_dtype(torch.float64)
        model64.load_state_dict(sd)
        assert torch.equal(model64._ref_edge_indexes, model._ref_edge_indexes)
        cap64 = capture(model64, lattice, zs, pos_batch, torch.float64)
        data["f64/forward"] = cap64["forward"]
        torch.set_default_dtype(torch.float64)
        data["f64/alpha"] = model64.calc_polarizabilities(pos_batch)
        torch.set_default_dtype(torch.float32)
        rel = np.abs(cap64["forward"] - cap["forward"]).max() / np.abs(cap64["forward"]).max()
        print(f"   f32 vs f64 forward max rel: {rel:.2e}")
        if extras is not None:
            extras(data, rng, ref, model, model64)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)


def phonon_md_extras(data, rng, ref, model, model64):
    """Phonon Raman tensors (f64 reference) + both spectra through the reference API."""
    n = ref.num_atoms
    m = 6
    disp = rng.normal(size=(m, n, 3)) @ np.linalg.inv(ref.lattice) * 0.3
    wn = np.linspace(80.0, 900.0, m)
    torch.set_default_dtype(torch.float64)
    spec = Phonons(ref.positions, wn, disp).get_raman_spectrum(model64)
    torch.set_default_dtype(torch.float32)
    data["ph/displacements"] = disp
    data["ph/wavenumbers"] = wn
    data["ph/raman_tensors"] = spec.raman_tensors
    w, i0 = spec.measure()
    data["ph/int_raw"] = i0
    w, i1 = spec.measure(laser_correction=True, laser_wavelength=532,
                         bose_einstein_correction=True, temperature=300)
    data["ph/int_corr"] = i1
    # MD: smooth synthetic trajectory, evaluated by the f32 reference model.
    s = 48
    tgrid = np.arange(s)[:, None, None]
    freq = rng.uniform(0.02, 0.2, (1, n, 3))
    phase = rng.uniform(0, 2 * np.pi, (1, n, 3))
    cart = 0.05 * np.cos(2 * np.pi * freq * tgrid + phase)
    traj_pos = ref.positions[None] + cart @ np.linalg.inv(ref.lattice)
    traj = Trajectory(traj_pos, 2.0)
    md = traj.get_raman_spectrum(model)
    data["md/positions"] = traj_pos
    data["md/timestep"] = np.float64(2.0)
    data["md/alpha_ts"] = md.polarizability_ts
    w, i0 = md.measure()
    data["md/wavenumbers"] = w
    data["md/int_raw"] = i0
    w, i1 = md.measure(laser_correction=True, laser_wavelength=532,
                       bose_einstein_correction=True, temperature=300)
    data["md/int_corr"] = i1


def train_step_extras(data, rng, ref, model, model64):
    """One optimisation step's forward/backward through the REFERENCE in train mode
    (batch-statistics BatchNorm): outputs, loss, every parameter gradient, running stats."""
    import copy
    m = copy.deepcopy(model)
    m.train()
    s = 4
    pos = data["pos_batch"][:s]
    target = torch.tensor(rng.normal(size=(s, 6)), dtype=torch.float32)
    lat = torch.tensor(ref.lattice, dtype=torch.float32).unsqueeze(0).expand(s, -1, -1)
    zz = torch.tensor(ref.atomic_numbers, dtype=torch.int).unsqueeze(0).expand(s, -1)
    out = m.forward(lat, zz, torch.tensor(pos, dtype=torch.float32))
    loss = torch.nn.MSELoss()(out, target)
    loss.backward()
    data["train/target"] = target.numpy()
    data["train/out"] = out.detach().numpy()
    data["train/loss"] = np.float64(loss.item())
    for k, p in m.named_parameters():
        data["train/grad/" + k] = p.grad.detach().numpy().copy()
    bn = m._to_polarizability_embedding[1]
    data["train/running_mean"] = bn.running_mean.numpy().copy()
    data["train/running_var"] = bn.running_var.numpy().copy()
    # the reference's dataset class on random polarizabilities
    from ramannoodle.dataset.torch import PolarizabilityDataset
    alpha = rng.normal(size=(7, 3, 3))
    alpha = alpha + np.swapaxes(alpha, 1, 2) + np.diag([30.0, 31.0, 29.0])
    ds = PolarizabilityDataset(ref.lattice, ref.atomic_numbers, data["pos_batch"][:1].repeat(7, 0), alpha)
    data["ds/alpha"] = alpha
    data["ds/scaled"] = ds.scaled_polarizabilities
    data["ds/mean"] = ds.mean_polarizability
    data["ds/std"] = ds.stddev_polarizability
    item = ds[3]
    data["ds/item3_target"] = item[3].numpy()
    ds.scale_polarizabilities(data["mean"], data["std"])
    data["ds/rescaled"] = ds.scaled_polarizabilities


def main():
    parity = dict(cutoff=2.0, fn=5, fe=14, passes=4, g0=0.0, g1=5.0)
    all_stages = ["unit", "dist", "node0", "edge0"] + [
        f"{k}{p}" for p in range(1, 5) for k in ("node", "edge")
    ] + ["pol_emb"]

    # A. the reference's own documented configuration on its own TiO2 cell.
    ref = rn.io.vasp.poscar.read_ref_structure("/root/reference/test/data/TiO2/POSCAR")
    make_case("tio2_notebook", ref.lattice, ref.positions, ref.atomic_numbers, parity,
              seed=101, style="notebook", s=4, keep=all_stages)

    # B. rocksalt 2x2x2 (N=64), cutoff 3.2 -> 18 neighbours; parity + perf widths.
    lat, pos, zs = rocksalt(2, 2, 2)
    hp = dict(parity, cutoff=3.2)
    make_case("rocksalt64_parity", lat, pos, zs, hp, seed=202, style="soft", s=3,
              keep=all_stages)
    hp = dict(cutoff=3.2, fn=64, fe=64, passes=4, g0=0.0, g1=5.0)
    make_case("rocksalt64_perf", lat, pos, zs, hp, seed=303, style="soft", s=2,
              keep=["node0", "node1", "node2", "node3", "node4", "pol_emb"])

    # C. triclinic, 3 species, ragged degrees, odd widths; + phonon / MD spectra.
    rng = np.random.default_rng(404)
    lat, pos, zs = triclinic(rng)
    hp = dict(cutoff=3.0, fn=8, fe=12, passes=2, g0=0.0, g1=4.0)
    keep = ["unit", "dist", "node0", "edge0", "node1", "edge1", "node2", "edge2", "pol_emb"]
    make_case("triclinic20", lat, pos, zs, hp, seed=404, style="soft", s=5, keep=keep,
              extras=phonon_md_extras)

    # C'. the same model through one training step of the reference (gradients of all weights)
    def both(data, rng_, ref_, model_, model64_):
        train_step_extras(data, rng_, ref_, model_, model64_)
    hp2 = dict(cutoff=3.0, fn=8, fe=12, passes=2, g0=0.0, g1=4.0)
    rng2 = np.random.default_rng(404)
    lat2, pos2, zs2 = triclinic(rng2)
    make_case("triclinic20_train", lat2, pos2, zs2, hp2, seed=404, style="soft", s=5, keep=[],
              extras=both)

    # D. sub-batch boundary: S=205 frames through calc_polarizabilities (100-frame chunks).
    lat, pos, zs = rocksalt(2, 2, 2)
    hp = dict(cutoff=2.5, fn=6, fe=10, passes=3, g0=0.0, g1=5.0)
    make_case("rocksalt64_s205", lat, pos, zs, hp, seed=505, style="soft", s=205,
              keep=[], with_f64=False)


if __name__ == "__main__":
    main()
