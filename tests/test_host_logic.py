"""Host-side logic of the product (no GPU): graph construction, API contract, spectra,
C-ABI symbol table."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from ramannoodle_amd import _lib
from ramannoodle_amd.dynamics import Phonons, Trajectory
from ramannoodle_amd.exceptions import DeviceError
from ramannoodle_amd.pmodel import PotGNN, graph as G
from ramannoodle_amd.pmodel import polarizability_tensors_to_vectors, polarizability_vectors_to_tensors
from ramannoodle_amd.spectrum import (MDRamanSpectrum, PhononRamanSpectrum, calc_signal_spectrum,
                                      convolve_spectrum, get_bose_einstein_correction,
                                      get_laser_correction)
from ramannoodle_amd.structure import ReferenceStructure, apply_pbc
from tests.conftest import ROOT, load_golden
from tests.helpers import product_model_from_golden


# ----------------------------------------------------------------------------- graph
def test_radius_graph_bit_exact(golden):
    name, g = golden
    edges = G.radius_graph_pbc(g["lattice"], g["positions"], float(g["hp"][0]))
    np.testing.assert_array_equal(edges, g["ref_edge_indexes"][1:])
    # sorted by (a, b), no self loops
    key = edges[0] * len(g["atomic_numbers"]) + edges[1]
    assert np.all(np.diff(key) > 0) and np.all(edges[0] != edges[1])


def test_reference_order_triplets_bit_exact(golden):
    name, g = golden
    trip = G.reference_order_triplets(g["ref_edge_indexes"][1:], len(g["atomic_numbers"]))
    for mine, key in zip(trip, ["i", "j", "idx_i", "idx_j", "idx_k", "slot5", "slot6"]):
        np.testing.assert_array_equal(mine, g["trip/" + key])


def test_atom_type_map(golden):
    name, g = golden
    np.testing.assert_array_equal(G.atom_type_map(g["atomic_numbers"]), g["atom_type_map"])


def test_margin_around_cutoff(golden):
    """Fixtures keep a >= 1e-4 A gap around the cutoff, so edge lists cannot flip on
    float32 round-off (SURVEY hard part 4)."""
    name, g = golden
    lat, pos = g["lattice"], g["positions"]
    d = pos[None] - pos[:, None]
    d = np.where(d % 1 > 0.5, d % 1 - 1, d % 1) @ lat
    dist = np.sqrt((d**2).sum(-1))
    np.fill_diagonal(dist, 1e9)
    assert np.abs(dist - float(g["hp"][0])).min() > 1e-4


# ----------------------------------------------------------------------------- model shell
def _ref():
    g = load_golden("triclinic20")
    return ReferenceStructure([int(z) for z in g["atomic_numbers"]], g["lattice"], g["positions"])


@pytest.mark.parametrize(
    "args, message",
    [
        ((-2.3, 5, 5, 5, 5, 5), "invalid cutoff: -2.3 <= 0"),
        ((2.3, -5, 5, 5, 5, 5), "invalid size_node_embedding: -5 <= 0"),
        ((2.3, 5, 0, 5, 5, 5), "invalid size_edge_embedding: 0 <= 0"),
        ((2.3, 5, 5, -1, 5, 5), "invalid num_message_passes: -1 <= 0"),
        ((2.3, 5, 5, 5, -0.5, 5), "invalid gaussian_filter_start: -0.5 < 0"),
        ((2.3, 5, 5, 5, 3, 2), "invalid gaussian_filter_end: 2 <= gaussian_filter_start"),
    ],
)
def test_constructor_validation_messages(args, message):
    """Same six messages the reference pins (test/tests/torch/test_gnn.py:196-301)."""
    with pytest.raises(ValueError, match=re.escape(message)):
        PotGNN(_ref(), *args, np.zeros((3, 3)), np.ones((3, 3)))


def test_constructor_shape_errors():
    with pytest.raises(ValueError, match=re.escape("mean_polarizability has wrong shape: (3,) != (3,3)")):
        PotGNN(_ref(), 3.0, 4, 4, 1, 0, 5, np.zeros(3), np.ones((3, 3)))
    with pytest.raises(TypeError, match="stddev_polarizability should have type ndarray, not list"):
        PotGNN(_ref(), 3.0, 4, 4, 1, 0, 5, np.zeros((3, 3)), [1, 2, 3])


def test_state_dict_layout_matches_reference(golden):
    name, g = golden
    model = product_model_from_golden(g)
    want = [(k[3:], g[k].shape) for k in g.files if k.startswith("sd/")]
    got = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert got == want
    assert model.gauss_coefficient == pytest.approx(float(g["gauss_coefficient"]), rel=0, abs=0)
    np.testing.assert_array_equal(model.ref_edge_indexes, g["ref_edge_indexes"])
    blob = model._weights_blob()
    cfg = _lib.Config(model.num_atoms, model.num_edges, model._num_atom_types, model._fn, model._fe,
                      model._passes, model.gauss_coefficient, 0, 0)
    assert blob.size == _lib.load().rn_potgnn_weight_count(ctypes.byref(cfg))


def test_same_seed_same_initial_weights_as_module_order():
    torch.manual_seed(3)
    a = PotGNN(_ref(), 3.0, 4, 6, 2, 0, 5, np.zeros((3, 3)), np.ones((3, 3))).state_dict()
    torch.manual_seed(3)
    b = PotGNN(_ref(), 3.0, 4, 6, 2, 0, 5, np.zeros((3, 3)), np.ones((3, 3))).state_dict()
    for k in a:
        assert torch.equal(a[k], b[k])


def test_wrong_positions_shape_is_value_error_before_any_device_work():
    model = product_model_from_golden(load_golden("triclinic20"))
    with pytest.raises(ValueError, match=re.escape("positions_batch has wrong shape: (2,5,3) != (_,20,3)")):
        model.calc_polarizabilities(np.zeros((2, 5, 3)))
    with pytest.raises(TypeError, match="positions_batch should have type ndarray, not list"):
        model.calc_polarizabilities([[1.0]])


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_device_fails_loudly():
    model = product_model_from_golden(load_golden("triclinic20"))
    with pytest.raises(DeviceError):
        model.calc_polarizabilities(load_golden("triclinic20")["pos_batch"])


def test_vector_tensor_maps():
    v = torch.arange(12.0).reshape(2, 6)
    t = polarizability_vectors_to_tensors(v)
    assert t.shape == (2, 3, 3) and torch.equal(t, t.transpose(1, 2))
    assert torch.equal(polarizability_tensors_to_vectors(t), v)
    with pytest.raises(ValueError):
        polarizability_vectors_to_tensors(torch.zeros(2, 5))


# ----------------------------------------------------------------------------- C ABI
def test_library_exports_every_declared_symbol():
    declared = set()
    for name in sorted(os.listdir(os.path.join(ROOT, "include"))):  # every header of the C ABI
        header = open(os.path.join(ROOT, "include", name)).read()
        header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
        declared |= set(re.findall(r"\b(rn_(?:potgnn|xdatcar|vasprun|host|md)_\w+)\s*\(", header))
    declared -= {"rn_potgnn_reduce_fn"}  # a function-pointer typedef, not an entry point
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.library_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in _lib.load().rn_potgnn_version()


def test_create_rejects_bad_arguments_without_touching_the_gpu():
    lib = _lib.load()
    cfg = _lib.Config(4, 2, 1, 200, 8, 1, -1.0, 0, 0)
    n = lib.rn_potgnn_weight_count(ctypes.byref(cfg))
    w = np.zeros(n, dtype=np.float32)
    ea = np.array([0, 1], dtype=np.int32)
    eb = np.array([1, 0], dtype=np.int32)
    ty = np.zeros(4, dtype=np.int32)
    lat = np.eye(3)
    h = ctypes.c_void_p()
    p = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
    rc = lib.rn_potgnn_create(ctypes.byref(cfg), p(ea), p(eb), p(ty), p(lat), p(w), n, p(lat), p(lat),
                              ctypes.byref(h))
    assert rc == _lib.RN_ERR_UNSUPPORTED and b"128" in lib.rn_potgnn_last_error(None)
    cfg.size_node_embedding = 8
    n = lib.rn_potgnn_weight_count(ctypes.byref(cfg))
    w = np.zeros(n, dtype=np.float32)
    eb_bad = np.array([1, 1], dtype=np.int32)  # edge (1,1) is a self loop
    rc = lib.rn_potgnn_create(ctypes.byref(cfg), p(ea), p(eb_bad), p(ty), p(lat), p(w), n, p(lat),
                              p(lat), ctypes.byref(h))
    assert rc == _lib.RN_ERR_INVALID_ARGUMENT
    ea_unsorted = np.array([1, 0], dtype=np.int32)
    rc = lib.rn_potgnn_create(ctypes.byref(cfg), p(ea_unsorted), p(ea), p(ty), p(lat), p(w), n,
                              p(lat), p(lat), ctypes.byref(h))
    assert rc == _lib.RN_ERR_INVALID_ARGUMENT and b"sorted" in lib.rn_potgnn_last_error(None)
    rc = lib.rn_potgnn_create(ctypes.byref(cfg), p(ea), p(eb), p(ty), p(lat), p(w), n - 1, p(lat),
                              p(lat), ctypes.byref(h))
    assert rc == _lib.RN_ERR_INVALID_ARGUMENT


def test_bench_refuses_to_run_without_gpus():
    """bench.py on a machine without a GPU: no CPU fallback, and ``--gpus N`` checks the device count
    before it starts any rank."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a GPU-less machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu"], capture_output=True, text=True,
                         timeout=300)
    assert one.returncode != 0 and "needs a GPU" in one.stderr
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--no-cpu"],
                         capture_output=True, text=True, timeout=300)
    assert two.returncode != 0 and "only 0 GPU(s) visible" in two.stderr
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    clash = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--no-cpu"], env=env,
                           capture_output=True, text=True, timeout=300)
    assert clash.returncode != 0 and "does not match WORLD_SIZE=2" in clash.stderr


def test_model_copies_and_pickles_carry_host_state_only():
    """``copy.deepcopy(model)`` / ``torch.save(model)`` (both appear around the reference's training
    loop) work on the device model: parameters are copied, the device handle is not."""
    import copy
    import io
    import torch
    from bench import rocksalt
    from ramannoodle_amd.pmodel import PotGNN
    from ramannoodle_amd.structure import ReferenceStructure
    lattice, ref, zs = rocksalt(1, 1, 1)
    model = PotGNN(ReferenceStructure(list(zs), lattice, ref), 3.0, 5, 14, 2, 0.0, 5.0, np.eye(3), np.ones((3, 3)))
    model._handle = ctypes.c_void_p(1234)  # what a live handle looks like to copy / pickle
    try:
        twin = copy.deepcopy(model)
        buffer = io.BytesIO()
        torch.save(model, buffer)
    finally:
        model._handle = None
    buffer.seek(0)
    loaded = torch.load(buffer, weights_only=False)
    for other in (twin, loaded):
        assert other._handle is None and other.num_edges == model.num_edges
        for (ka, va), (kb, vb) in zip(model.state_dict().items(), other.state_dict().items()):
            assert ka == kb and torch.equal(va, vb)
        assert next(other.parameters()) is not next(model.parameters())
        assert isinstance(next(other.parameters()), torch.nn.Parameter)
        assert isinstance(other, torch.nn.Module)


def test_documented_size_limits_raise_not_implemented():
    """The two limits of the device model (DESIGN.md "Limits"): embedding sizes above 128, and
    more outgoing edges per atom than one LDS tile holds (149 at Fe = 64; the reference's
    ``radius_graph_pbc`` keeps one edge per ordered atom pair, so this needs > 149 atoms inside
    the cutoff sphere).  Both are refused at model creation with NotImplementedError, before
    any device work."""
    from bench import rocksalt
    from ramannoodle_amd.pmodel import PotGNN
    from ramannoodle_amd.structure import ReferenceStructure
    lattice, ref, zs = rocksalt(1, 1, 1)
    small = ReferenceStructure(list(zs), lattice, ref)
    unit = (np.eye(3), np.ones((3, 3)))
    for fn, fe in ((129, 8), (8, 129)):
        model = PotGNN(small, 3.0, fn, fe, 1, 0.0, 5.0, *unit)
        with pytest.raises(NotImplementedError, match="embedding sizes above 128 are not supported"):
            model.calc_polarizabilities(ref[None])
    lattice, ref, zs = rocksalt(4, 4, 2)
    dense = PotGNN(ReferenceStructure(list(zs), lattice, ref), 7.9, 64, 64, 1, 0.0, 5.0, *unit)
    assert dense.num_edges / dense.num_atoms > 149
    with pytest.raises(NotImplementedError,
                       match=r"outgoing edges; more than 149 per atom is unsupported for size_edge_embedding=64"):
        dense.calc_polarizabilities(ref[None])
    # the same graph is accepted at a width whose rows fit (590 per atom at Fe <= 16): creation then
    # proceeds to the device probe, which is what fails on a machine without a GPU
    import torch
    if not torch.cuda.is_available():
        from ramannoodle_amd.exceptions import DeviceError
        ok = PotGNN(ReferenceStructure(list(zs), lattice, ref), 7.9, 8, 16, 1, 0.0, 5.0, *unit)
        with pytest.raises(DeviceError, match="no usable HIP device"):
            ok.calc_polarizabilities(ref[None])


# ----------------------------------------------------------------------------- spectra
def test_phonon_spectrum_matches_reference():
    g = load_golden("triclinic20")
    spec = PhononRamanSpectrum(g["ph/wavenumbers"], g["ph/raman_tensors"])
    w, i = spec.measure()
    np.testing.assert_array_equal(w, g["ph/wavenumbers"])
    np.testing.assert_allclose(i, g["ph/int_raw"], rtol=1e-13)
    w, i = spec.measure(laser_correction=True, laser_wavelength=532,
                        bose_einstein_correction=True, temperature=300)
    np.testing.assert_allclose(i, g["ph/int_corr"], rtol=1e-13)
    with pytest.raises(NotImplementedError):
        spec.measure(orientation=np.eye(3))


def test_md_spectrum_matches_reference():
    g = load_golden("triclinic20")
    spec = MDRamanSpectrum(g["md/alpha_ts"], float(g["md/timestep"]))
    w, i = spec.measure()
    np.testing.assert_allclose(w, g["md/wavenumbers"], rtol=1e-14)
    np.testing.assert_allclose(i, g["md/int_raw"], rtol=1e-10, atol=1e-12 * np.abs(g["md/int_raw"]).max())
    w, i = spec.measure(laser_correction=True, laser_wavelength=532,
                        bose_einstein_correction=True, temperature=300)
    np.testing.assert_allclose(i, g["md/int_corr"], rtol=1e-10, atol=1e-12 * np.abs(g["md/int_corr"]).max())
    s = g["md/alpha_ts"].shape[0]
    assert w.shape == (int(np.ceil((s - 1) / 2)) - 1 + ((s - 1) % 2 == 0) * 0,) or w.size > 0


def test_signal_spectrum_length():
    """Output length ceil(S/2) (test/tests/test_trajectory_spectrum.py:18-32)."""
    for s in (40, 51):
        w, i = calc_signal_spectrum(np.random.default_rng(0).normal(size=s), 1.0)
        assert w.shape == i.shape == (int(np.ceil(s / 2)),)


def test_corrections_and_convolution_errors():
    w = np.array([100.0, 200.0])
    with pytest.raises(ValueError, match="invalid temperature: -1 <= 0"):
        get_bose_einstein_correction(w, -1)
    with pytest.raises(TypeError, match="temperature should have type float, not list"):
        get_bose_einstein_correction(w, [])
    with pytest.raises(ValueError, match="invalid laser_wavenumber: 0 <= 0"):
        get_laser_correction(w, 0)
    with pytest.raises(ValueError, match="unsupported convolution type: triangle"):
        convolve_spectrum(w, w, "triangle")
    ow, oi = convolve_spectrum(w, np.array([1.0, 2.0]), "gaussian", 5)
    assert oi.sum() * (ow[1] - ow[0]) == pytest.approx(3.0, rel=1e-3)
    ow, oi = convolve_spectrum(w, np.array([1.0, 2.0]), "lorentzian", 5, out_wavenumbers=np.linspace(0, 300, 7))
    assert ow.shape == oi.shape == (7,)


def test_convolve_spectrum_against_the_reference_tests_known_answers():
    """The known-answer vectors of the reference's own ``test_convolve_intensities``
    (``test/tests/test_phonon_spectrum.py:404-447``, data under tests/golden/ref_spectrum), with
    that test's tolerance (``np.allclose``)."""
    base = os.path.join(os.path.dirname(__file__), "golden", "ref_spectrum")
    known = np.load(os.path.join(base, "known_spectrum.npz"))
    for function in ("gaussian", "lorentzian"):
        expected = np.load(os.path.join(base, f"known_{function}_spectrum.npz"))
        w, i = convolve_spectrum(known["wavenumbers"], known["intensities"], function)
        assert np.allclose(w, expected["wavenumbers"])
        assert np.allclose(i, expected["intensities"])
        np.testing.assert_allclose(i, expected["intensities"], rtol=1e-9, atol=1e-12 * np.abs(expected["intensities"]).max())


@pytest.mark.parametrize("wavenumbers,intensities,function,width,out_wavenumbers,exception_type,in_reason", [
    # the table of the reference's test_convolve_intensities_exception (test_phonon_spectrum.py:450-541)
    (np.array([1, 2, 3]), [0, 3, 0], "gaussian", 5, None, TypeError, "intensities should have type ndarray, not list"),
    ([1, 2, 3], np.array([0, 3, 0]), "gaussian", 5, None, TypeError, "wavenumbers should have type ndarray, not list"),
    (np.array([1, 2, 3, 4]), np.array([0, 3, 0]), "gaussian", 5, None, ValueError,
     "intensities has wrong shape: (3,) != (4,)"),
    (np.array([[1, 2, 3, 5], [2, 3, 4, 5]]), np.array([0, 3, 0, 4]), "gaussian", 5, None, ValueError,
     "wavenumbers has wrong shape: (2,4) != (_,)"),
    (np.array([1, 2, 3]), np.array([0, 3, 0]), "smoother", 5, None, ValueError, "unsupported convolution type: smoother"),
    (np.array([1, 2, 3]), np.array([0, 3, 0]), "gaussian", -4, None, ValueError, "invalid width: -4 <= 0"),
    (np.array([1, 2, 3]), np.array([0, 3, 0]), "gaussian", "not_a_int", None, TypeError,
     "width should have type float, not str"),
    (np.array([1, 2, 3]), np.array([0, 3, 0]), "gaussian", 5.0, np.array(([1, 2], [1, 2])), ValueError,
     "out_wavenumbers has wrong shape: (2,2) != (_,)"),
])
def test_convolve_spectrum_exceptions_as_the_reference(wavenumbers, intensities, function, width, out_wavenumbers,
                                                       exception_type, in_reason):
    import re
    with pytest.raises(exception_type, match=re.escape(in_reason)):
        convolve_spectrum(wavenumbers, intensities, function, width, out_wavenumbers)


@pytest.mark.parametrize("function,argument,exception_type,in_reason", [
    # test_get_bose_einstein_correction_exception / test_get_laser_correction (test_phonon_spectrum.py:544-618)
    (get_bose_einstein_correction, 300, TypeError, "wavenumbers should have type ndarray, not list"),
    (get_bose_einstein_correction, -300, ValueError, "invalid temperature: -300 <= 0"),
    (get_bose_einstein_correction, "string temperature", TypeError, "temperature should have type float, not str"),
    (get_laser_correction, 600, TypeError, "wavenumbers should have type ndarray, not list"),
    (get_laser_correction, -600, ValueError, "invalid laser_wavenumber: -600 <= 0"),
    (get_laser_correction, "string wavelength", TypeError, "laser_wavenumber should have type float, not str"),
])
def test_correction_exceptions_as_the_reference(function, argument, exception_type, in_reason):
    import re
    with pytest.raises(exception_type, match=re.escape(in_reason)):
        function([1, 2, 3], argument)


# ----------------------------------------------------------------------------- dynamics
class _Raises:
    def calc_polarizabilities(self, positions_batch):
        raise ValueError("positions_batch has wrong shape")


class _Linear:
    """Deterministic stand-in model: alpha = M . mean displacement (host only)."""

    def __init__(self, n):
        rng = np.random.default_rng(5)
        self.w = rng.normal(size=(n * 3, 6))

    def calc_polarizabilities(self, positions_batch):
        v = positions_batch.reshape(positions_batch.shape[0], -1) @ self.w
        return v[:, [[0, 3, 4], [3, 1, 5], [4, 5, 2]]]


def test_dynamics_constructor_and_incompatibility_errors():
    ref = np.zeros((4, 3))
    with pytest.raises(ValueError, match=re.escape("displacements has wrong shape: (2,3,3) != (2,4,3)")):
        Phonons(ref, np.array([1.0, 2.0]), np.zeros((2, 3, 3)))
    ph = Phonons(ref, np.array([1.0, 2.0]), np.zeros((2, 4, 3)))
    with pytest.raises(ValueError, match="polarizability_model and phonons are incompatible"):
        ph.get_raman_spectrum(_Raises())
    with pytest.raises(ValueError, match="timestep must be positive"):
        Trajectory(np.zeros((3, 4, 3)), 0)
    with pytest.raises(TypeError, match="timestep should have type float, not list"):
        Trajectory(np.zeros((3, 4, 3)), [])
    tr = Trajectory(np.zeros((3, 4, 3)) + 1.25, 1.0)
    assert np.allclose(tr.positions_ts, 0.25) and len(tr) == 3 and tr[1].shape == (4, 3)
    with pytest.raises(IndexError, match="trajectory index out of bounds"):
        tr[7]
    with pytest.raises(ValueError, match="polarizability_model and trajectory are incompatible"):
        tr.get_raman_spectrum(_Raises())


def test_phonon_finite_difference_divides_by_delta_not_two_delta():
    n = 4
    model = _Linear(n)
    rng = np.random.default_rng(1)
    ref, disp = rng.uniform(size=(n, 3)), rng.normal(size=(3, n, 3))
    spec = Phonons(ref, np.array([10.0, 20.0, 30.0]), disp).get_raman_spectrum(model)
    expect = 2.0 * (disp.reshape(3, -1) @ model.w)[:, [[0, 3, 4], [3, 1, 5], [4, 5, 2]]]
    np.testing.assert_allclose(spec.raman_tensors, expect, rtol=1e-9, atol=1e-9)


def test_apply_pbc():
    x = np.array([[-0.25, 1.5, 0.0]])
    np.testing.assert_allclose(apply_pbc(x), [[0.75, 0.5, 0.0]])
    with pytest.raises(TypeError):
        apply_pbc("nope")


def test_parameters_apply_and_modes():
    """torch-facing surface of the model shell: Parameters for torch.optim, the notebook's
    `model.apply(init_fn)` idiom (machine-learning.ipynb:198-206), train/eval flags."""
    model = product_model_from_golden(load_golden("triclinic20"))
    assert isinstance(model, torch.nn.Module)  # like the reference's (_gnn.py:418-421)
    params = list(model.parameters())
    assert all(isinstance(p, torch.nn.Parameter) for p in params)
    assert sum(p.numel() for p in params) == 4500  # the reference's parameter count for this model
    names = [k for k, _ in model.named_parameters()]
    assert "_edge_embedding.offset" not in names and "_to_polarizability_embedding.1.running_mean" not in names

    def init_biases(m):
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.uniform_(m.bias, -0.5, 0.5)

    def init_weights(m):
        if isinstance(m, torch.nn.Linear) or isinstance(m, torch.nn.Embedding):
            torch.nn.init.normal_(m.weight, mean=0, std=1)

    before = {k: v.clone() for k, v in model.state_dict().items()}  # (torch's state_dict() aliases the parameters)
    torch.manual_seed(1)
    model.apply(init_biases).apply(init_weights)
    after = {k: v.clone() for k, v in model.state_dict().items()}
    assert not torch.equal(before["_node_embedding.0.weight"], after["_node_embedding.0.weight"])
    assert not torch.equal(before["_edge_blocks.1.c3_linear.bias"], after["_edge_blocks.1.c3_linear.bias"])
    assert torch.equal(before["_node_blocks.0.c1_norm.weight"], after["_node_blocks.0.c1_norm.weight"])
    assert float(after["_to_polarizability_embedding.5.bias"].abs().max()) <= 0.5
    # load_state_dict keeps the Parameter objects (optimisers stay valid)
    model.load_state_dict(before)
    assert all(a is b for a, b in zip(params, model.parameters()))
    assert model.training and not model.eval().training and model.train().training
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    next(model.parameters()).grad = torch.ones_like(next(model.parameters()))
    opt.step()
    assert not torch.equal(model.state_dict()["_node_embedding.0.weight"], before["_node_embedding.0.weight"])


def test_reset_parameters():
    """(reference: test/tests/torch/test_gnn.py:116-120)"""
    model = product_model_from_golden(load_golden("triclinic20"))
    before = {k: v.clone() for k, v in model.state_dict().items()}
    params = list(model.parameters())
    torch.manual_seed(11)
    model.reset_parameters()
    after = model.state_dict()
    assert not torch.equal(before["_node_blocks.0.c1_linear.weight"], after["_node_blocks.0.c1_linear.weight"])
    assert torch.equal(after["_node_blocks.1.final_norm.weight"], torch.ones(8))
    assert torch.equal(after["_to_polarizability_embedding.1.running_var"], torch.ones(12))
    assert all(a is b for a, b in zip(params, model.parameters()))
    torch.manual_seed(11)
    from ramannoodle_amd.pmodel import PotGNN
    g = load_golden("triclinic20")
    fresh = PotGNN(_ref(), 3.0, 8, 12, 2, 0.0, 4.0, g["mean"], g["std"]).state_dict()
    for k in after:
        assert torch.equal(after[k], fresh[k]), k


def test_config1_plumbing_linear_model_phonon_spectrum():
    """BASELINE config 1 (CPU plumbing, no GPU): an 8-atom rocksalt cell, a seeded linear
    ``PolarizabilityModel`` (what an order-1 P1 InterpolationModel evaluates,
    ``pmodel/_interpolation.py:239-244``) and 24 modes through ``Phonons.get_raman_spectrum`` ->
    ``PhononRamanSpectrum.measure``, against what the reference's own classes produced
    (tests/golden/make_golden_r2.py).  Exercises the generic per-mode loop of
    ``dynamics/_phonon.py:93-106`` (the model has no ``calc_raman_tensors``)."""
    from ramannoodle_amd.abstract import PolarizabilityModel
    from ramannoodle_amd.dynamics import Phonons

    g = load_golden("config1_plumbing")

    class LinearModel(PolarizabilityModel):
        def __init__(self):
            self.calls = 0

        def calc_polarizabilities(self, positions_batch):
            self.calls += 1
            d = (positions_batch - g["positions"][None]).reshape(positions_batch.shape[0], -1)
            return g["alpha0"][None] + np.einsum("sk,kij->sij", d, g["coeff"])

    model = LinearModel()
    spectrum = Phonons(g["positions"], g["wavenumbers"], g["displacements"]).get_raman_spectrum(model)
    assert model.calls == 2 * 24  # +-delta per mode, batch of one each (the reference's loop)
    np.testing.assert_allclose(spectrum.raman_tensors, g["raman_tensors"], rtol=0,
                               atol=1e-12 * np.abs(g["raman_tensors"]).max())
    w, i = spectrum.measure()
    np.testing.assert_array_equal(w, g["out_wavenumbers"])
    np.testing.assert_allclose(i, g["int_raw"], rtol=1e-10)
    w, i = spectrum.measure(laser_correction=True, laser_wavelength=522, bose_einstein_correction=True,
                            temperature=300)
    np.testing.assert_allclose(i, g["int_corr"], rtol=1e-10)
    # a model that rejects the shape surfaces as the reference's "incompatible" ValueError
    class Wrong(PolarizabilityModel):
        def calc_polarizabilities(self, positions_batch):
            raise ValueError("positions_batch has wrong shape")

    with pytest.raises(ValueError, match="incompatible"):
        Phonons(g["positions"], g["wavenumbers"], g["displacements"]).get_raman_spectrum(Wrong())


def test_role_split_producers_leave_in_flight_registers_alone():
    """The producers of the role-specialised EdgeBlock (``csrc/kernels_edge_ps.hip``) load their node terms in inline assembly
    and wait for them with a hand-counted ``s_waitcnt vmcnt(5)`` behind the first product.  ``tools/check_ps_isa.py`` compiles
    the file and checks in the ISA of every instantiation that nothing reads, copies or spills those registers before the
    wait (a tied-operand copy in front of a wait on a second code path did exactly that once)."""
    import shutil
    import subprocess
    import sys
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "check_ps_isa.py")], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 violation(s)" in out.stdout and "0 in-flight loads" not in out.stdout


def _tile_rows(atoms, degree):
    """First / end source row of every destination of a tile of `atoms` atoms with `degree` in- and out-edges each."""
    rb = np.repeat(np.arange(atoms) * degree, degree).astype(np.int32)
    return rb, (rb + degree).astype(np.int32)


def test_role_split_schedule_check_on_the_host():
    """The schedule check behind the role-specialised EdgeBlock (``rn_potgnn_debug_ps_schedule``, host-only: it is what
    decides at model creation whether a graph takes ``csrc/kernels_edge_ps.hip``, and with how much lookahead).  On the
    benchmark cell's tile -- 8 atoms with 18 in- and out-edges, 9 rounds of 16 destinations -- the ring must
    hold 5 / 6 / 7 tiles for 1 / 2 / 3 rounds still reading when a step rewrites it, a round's window spans at most 3
    tiles, and a 47-neighbour atom (TiO2 at 5 A) does not fit the 8-tile ring at all."""
    import ctypes as C
    from ramannoodle_amd import _lib
    lib = _lib.load()

    def check(rb, re, back, ring):
        window = C.c_int32(-1)
        rc = lib.rn_potgnn_debug_ps_schedule(C.c_void_p(rb.ctypes.data), C.c_void_p(re.ctypes.data), len(rb), back, ring,
                                             C.byref(window))
        assert rc in (0, 1), rc
        return bool(rc), window.value

    rb, re = _tile_rows(8, 18)
    need = {}
    for back in (1, 2, 3):
        need[back] = next(ring for ring in range(1, 17) if check(rb, re, back, ring)[0])
        assert all(check(rb, re, back, ring)[0] for ring in range(need[back], 17))  # (monotone in the capacity)
    assert need == {1: 5, 2: 6, 3: 7}
    assert check(rb, re, 3, 8) == (True, 3) and check(rb, re, 3, 7) == (True, 3) and not check(rb, re, 3, 6)[0]
    # fewer neighbours (a round of 16 destinations then spans three or four atoms); a ragged tile is decided without a device too
    assert check(*_tile_rows(16, 6), 3, 8) == (True, 3)
    rng = np.random.default_rng(3)
    deg = rng.integers(1, 12, size=10)
    start = np.concatenate([[0], np.cumsum(deg)])
    rb_r = np.repeat(start[:-1], deg).astype(np.int32)
    re_r = np.repeat(start[1:], deg).astype(np.int32)
    assert check(rb_r, re_r, 2, 8)[0]
    # TiO2 at 5 A: 47 out-edges per atom -- a round's window spans seven tiles, two rounds in flight need nine
    assert check(*_tile_rows(4, 47), 2, 8) == (False, 7) and check(*_tile_rows(4, 47), 2, 9)[0]
    assert lib.rn_potgnn_debug_ps_schedule(None, None, 0, 2, 8, None) < 0


def test_committed_profile_records_are_fresh():
    """`bench.py` prints `roofline.traffic` / `issue_frac` from PMC / SQ passes committed under profiles/rNN/ -- next to a
    FRESH timing.  Every record carries the git blob hash of the kernel source it was taken on and is refused when that file
    has changed since; this test fails as soon as a kernel edit leaves the newest records behind, so
    that the profile is re-taken (tools/profile.sh + tools/make_profile_json.py) before the change is committed."""
    import json
    import bench
    shapes = {"perf": (256, 4608, 64, 64), "parity": (256, 4608, 5, 14)}
    for name, shape in (("edge_ps_traffic.json", "perf"), ("edge_ps_issue.json", "perf"), ("node_atom_traffic.json", "perf"),
                        ("edge_narrow_traffic.json", "parity"), ("edge_narrow_issue.json", "parity"),
                        ("node_narrow_traffic.json", "parity")):
        rec = bench.committed_profile(name, *shapes[shape])
        assert rec is not None, f"{name}: no record of this shape whose kernel source is the one in this tree"
        assert rec["_path"].startswith("profiles/r06"), rec["_path"]
    # a record without a stamp, or with a foreign one, is refused
    stale = os.path.join(bench.ROOT, "profiles", "r05", "edge_ps_traffic.json")
    assert "kernel_source" not in json.load(open(stale))
    older = bench.PROFILE_ROUNDS
    try:
        bench.PROFILE_ROUNDS = ("r05",)
        assert bench.committed_profile("edge_ps_traffic.json", 256, 4608, 64, 64) is None
    finally:
        bench.PROFILE_ROUNDS = older
