"""Pin the CPU oracle against fixtures produced by executing the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import potgnn_oracle as O


def test_graph_and_triplet_indices_bit_exact(golden):
    name, g = golden
    m = O.model_from_arrays(g)
    edges, trip, tmap = O.build_topology(g["lattice"], g["positions"], g["atomic_numbers"],
                                         float(g["hp"][0]))
    assert torch.equal(edges, m.edges)
    for mine, ref in zip(trip, m.trip):
        assert torch.equal(mine, ref)
    assert torch.equal(tmap.int(), m.atom_type_map.int())


def test_forward_stages_match_reference(golden):
    name, g = golden
    m = O.model_from_arrays(g)
    stages = {}
    out = O.forward(m, g["pos_batch"], faithful=True, stages=stages)
    np.testing.assert_allclose(out.numpy(), g["f32/forward"], rtol=0, atol=2e-6)
    for key in g.files:
        if key.startswith("f32/") and key[4:] in stages:
            np.testing.assert_allclose(stages[key[4:]].numpy(), g[key], rtol=0, atol=3e-6,
                                       err_msg=f"{name}:{key}")


def test_lean_variant_equals_faithful(golden):
    name, g = golden
    m = O.model_from_arrays(g)
    pos = g["pos_batch"][:3]
    a = O.forward(m, pos, faithful=True).numpy()
    b = O.forward(m, pos, faithful=False).numpy()
    np.testing.assert_allclose(a, b, rtol=0, atol=5e-6)


def test_calc_polarizabilities_f32_and_f64(golden):
    name, g = golden
    m = O.model_from_arrays(g)
    alpha = O.calc_polarizabilities(m, g["pos_batch"])
    np.testing.assert_allclose(alpha, g["f32/alpha"], rtol=1e-6, atol=1e-6)
    assert alpha.dtype == np.float64 and alpha.shape == (g["pos_batch"].shape[0], 3, 3)
    np.testing.assert_array_equal(alpha, np.swapaxes(alpha, 1, 2))
    if "f64/alpha" in g.files:
        a64 = O.calc_polarizabilities(m.to(torch.float64), g["pos_batch"][:2])
        np.testing.assert_allclose(a64, g["f64/alpha"][:2], rtol=1e-7, atol=1e-7)


def test_wrong_shape_is_value_error():
    g = np.load("tests/golden/triclinic20.npz")
    m = O.model_from_arrays(g)
    with pytest.raises(ValueError):
        O.calc_polarizabilities(m, np.zeros((2, 5, 3)))


def test_phonon_raman_tensors_f64():
    g = np.load("tests/golden/triclinic20.npz")
    m = O.model_from_arrays(g).to(torch.float64)
    r = O.raman_tensors_fd(m, g["positions"], g["ph/displacements"])
    scale = np.abs(g["ph/raman_tensors"]).max()
    np.testing.assert_allclose(r, g["ph/raman_tensors"], rtol=0, atol=2e-5 * scale)


def test_forward_with_per_sample_lattices_matches_reference():
    """``forward(lattice[S,3,3], ...)`` uses each sample's own lattice in the geometry
    (_gnn.py:603-611); fixture: the reference's forward on six strained cells, float32 and
    float64 (tests/golden/make_golden_r2.py)."""
    from tests.conftest import load_golden
    g, r = load_golden("triclinic20"), load_golden("triclinic20_r2")
    m = O.model_from_arrays(g)
    out = O.forward(m, r["lat/positions"], faithful=True, lattices=r["lat/lattices"]).numpy()
    np.testing.assert_array_equal(out, r["lat/forward"])
    sane = O.forward(m, r["lat/positions"], faithful=False, lattices=r["lat/lattices"]).numpy()
    np.testing.assert_allclose(sane, r["lat/forward"], rtol=0, atol=2e-7)
    m64 = O.model_from_arrays(g).to(torch.float64)
    # (a float64 reference model derives its Gaussian coefficient from a float64 linspace, _gnn.py:63-64)
    m64.coefficient = -0.5 / ((float(g["hp"][5]) - float(g["hp"][4])) / (int(g["hp"][2]) - 1)) ** 2
    out64 = O.forward(m64, r["lat/positions"], faithful=False, lattices=r["lat/lattices"]).numpy()
    np.testing.assert_allclose(out64, r["lat/forward64"], rtol=0, atol=1e-13)
    # sample 0 carries the reference lattice: same as the plain call
    plain = O.forward(m, r["lat/positions"][:1], faithful=True).numpy()
    np.testing.assert_array_equal(plain[0], r["lat/forward"][0])


def test_forward_with_per_sample_species_matches_reference():
    """``forward(lattice, atomic_numbers[S,N], positions)`` with species that differ between samples
    (``_convert_to_atom_type``, ``_gnn.py:541-557``): the oracle's ``atomic_numbers`` argument against the reference's
    float32 and float64 outputs (fixture ``triclinic20_r3``: atoms swapped, one species replaced, strained lattices).
    This pins what the training-gradient test with per-sample inputs differentiates."""
    from tests.conftest import load_golden
    g, r = load_golden("triclinic20"), load_golden("triclinic20_r3")
    m = O.model_from_arrays(g)
    out = O.forward(m, r["zs/positions"], faithful=True, lattices=r["zs/lattices"], atomic_numbers=r["zs/atomic_numbers"]).numpy()
    np.testing.assert_allclose(out, r["zs/forward"], rtol=0, atol=3e-7)
    m64 = O.model_from_arrays(g).to(torch.float64)
    m64.coefficient = -0.5 / ((float(g["hp"][5]) - float(g["hp"][4])) / (int(g["hp"][2]) - 1)) ** 2
    out64 = O.forward(m64, r["zs/positions"], faithful=False, lattices=r["zs/lattices"],
                      atomic_numbers=r["zs/atomic_numbers"]).numpy()
    np.testing.assert_allclose(out64, r["zs/forward64"], rtol=0, atol=1e-12)


def test_forward_on_positions_far_outside_the_unit_cell():
    """The reference's own batch test feeds ``forward`` positions drawn from N(0,1)
    (``test/tests/torch/test_gnn.py:83-113``): fractional coordinates up to +-3, negative ones
    included.  Fixture: the reference's float32 and float64 outputs for such a batch
    (tests/golden/make_golden_r2.py ``randn``)."""
    from tests.conftest import load_golden
    g, r = load_golden("triclinic20"), load_golden("triclinic20_randn")
    assert np.abs(r["positions"]).max() > 2.5 and r["positions"].min() < -2.0
    m = O.model_from_arrays(g)
    pos = r["positions"].astype(np.float64)
    np.testing.assert_array_equal(O.forward(m, pos, faithful=True).numpy(), r["forward"])
    np.testing.assert_allclose(O.forward(m, pos, faithful=False).numpy(), r["forward"], rtol=0, atol=3e-7)
    for i in range(pos.shape[0]):  # batch-size independence, the property the reference's test asserts
        np.testing.assert_allclose(O.forward(m, pos[i:i + 1], faithful=True).numpy()[0], r["forward"][i],
                                   rtol=0, atol=1e-5)



def test_triplet_fixture_digests():
    """The one unpinned point of the oracle: the ORDER of edge triplets is a restatement of
    ``torch_geometric...dimenet.triplets``.  ``tools/check_triplets_with_pyg.py`` checks the fixtures' seven ``trip/*`` arrays
    against a real torch_geometric wherever one is installed; this test keeps what that script would check from drifting:
    the SHA-256 of every fixture's arrays is the committed one, and the oracle's ``triplets`` reproduces the arrays."""
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_triplets_with_pyg", os.path.join(root, "tools", "check_triplets_with_pyg.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    recorded = json.load(open(os.path.join(root, "tests", "golden", "triplet_hashes.json")))
    seen = 0
    for name, g in tool.fixtures():
        assert recorded.get(name) == tool.digest(g), name
        got = O.triplets(torch.as_tensor(g["ref_edge_indexes"][[1, 2]], dtype=torch.long), int(g["positions"].shape[0]))
        for key, arr in zip(tool.KEYS, got):
            np.testing.assert_array_equal(arr.numpy(), g["trip/" + key], err_msg=f"{name} trip/{key}")
        seen += 1
    assert seen == len(recorded) >= 5
