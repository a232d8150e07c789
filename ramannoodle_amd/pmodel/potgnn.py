"""PotGNN polarizability model evaluated by hand-written gfx950 kernels.

Host-side mirror of ``ramannoodle.pmodel.torch.PotGNN`` (``ramannoodle/pmodel/torch/
_gnn.py:418-721``) for the evaluation path: same constructor signature and error
messages, same ``state_dict`` keys, ``forward`` / ``calc_polarizabilities`` with the same
argument meaning.  All arithmetic on positions happens on the device through the C ABI
in ``include/rn_potgnn.h``; PyTorch is used only to hold parameter tensors and device
buffers.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np
import torch
from numpy.typing import NDArray

from ramannoodle_amd import _lib
from ramannoodle_amd.abstract import PolarizabilityModel
from ramannoodle_amd.constants import RAMAN_TENSOR_CENTRAL_DIFFERENCE
from ramannoodle_amd.exceptions import verify_ndarray_shape
from ramannoodle_amd.pmodel import graph as graph_utils

_VEC_TO_TENSOR = np.array([[0, 3, 4], [3, 1, 5], [4, 5, 2]])  # dataset/torch/utils.py:30-37


def polarizability_vectors_to_tensors(vectors):
    """``[S,6] (xx,yy,zz,xy,xz,yz) -> [S,3,3]`` (``dataset/torch/utils.py:20-41``)."""
    if vectors.ndim != 2 or vectors.shape[1] != 6:
        raise ValueError(f"polarizability_vectors has wrong size: {list(vectors.shape)} != [_,6]")
    return vectors[:, _VEC_TO_TENSOR]


def polarizability_tensors_to_vectors(tensors):
    """``[S,3,3] -> [S,6]`` (``dataset/torch/utils.py:44-60``)."""
    if tensors.ndim != 3 or tuple(tensors.shape[1:]) != (3, 3):
        raise ValueError(f"polarizability_tensors has wrong size: {list(tensors.shape)} != [_,3,3]")
    return tensors[:, [0, 1, 2, 0, 0, 1], [0, 1, 2, 1, 2, 2]]


def _wants_float64(dtype) -> bool:
    """Evaluation dtype selector: ``None`` = torch's default dtype (``_gnn.py:705``)."""
    if dtype is None:
        dtype = torch.get_default_dtype()
    if dtype in (torch.float64, np.float64, float, "float64", "f64"):
        return True
    if dtype in (torch.float32, np.float32, "float32", "f32"):
        return False
    raise ValueError(f"unsupported evaluation dtype: {dtype}")


_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_void_p)


def _ptr(array) -> C.c_void_p:
    return C.c_void_p(array.ctypes.data)


class _Activation(torch.nn.Module):
    """Placeholder for the reference's parameter-free ``ShiftedSoftplus`` entries of its ``Sequential`` containers
    (``_gnn.py:508-514, 532-539``): keeps the children's indices -- and with them the ``state_dict`` keys -- the
    reference's.  The arithmetic happens in the kernels."""

    def forward(self, x):  # pylint: disable=missing-function-docstring
        return torch.nn.functional.softplus(x) - 0.6931471805599453


class _GaussianFilter(torch.nn.Module):
    """Holder of the ``offset`` buffer (``_gnn.py:53-66``)."""

    def __init__(self, start: float, end: float, steps: int):
        super().__init__()
        self.register_buffer("offset", torch.linspace(start, end, steps, dtype=torch.float32))


class _NodeBlockParams(torch.nn.Module):
    """Parameters of ``_NodeBlock`` under the reference's names, in its order (``_gnn.py:106-120``)."""

    def __init__(self, fn: int, fe: int):
        super().__init__()
        self.c1_linear = torch.nn.Linear(fn + fe, 2 * fn)
        self.c1_norm = torch.nn.LayerNorm(2 * fn)
        self.final_norm = torch.nn.LayerNorm(fn)


class _EdgeBlockParams(torch.nn.Module):
    """Parameters of ``_EdgeBlock`` under the reference's names, in its order (``_gnn.py:180-205``)."""

    def __init__(self, fn: int, fe: int):
        super().__init__()
        self.c2_linear = torch.nn.Linear(fn, 2 * fe)
        self.c3_linear = torch.nn.Linear(3 * fn + 2 * fe, 2 * fe)
        self.c2_norm_1 = torch.nn.LayerNorm(2 * fe)
        self.c3_norm_1 = torch.nn.LayerNorm(2 * fe)
        self.c2_norm_2 = torch.nn.LayerNorm(fe)
        self.c3_norm_2 = torch.nn.LayerNorm(fe)


class PotGNN(torch.nn.Module, PolarizabilityModel):  # pylint: disable=too-many-instance-attributes
    """POlarizability Tensor Graph Neural Network, device evaluation.

    A ``torch.nn.Module`` like the reference's (``_gnn.py:418-421``): its children are the reference's containers
    (``_node_embedding``, ``_edge_embedding``, ``_node_blocks``, ``_edge_blocks``, ``_to_polarizability_embedding``)
    holding real ``Linear`` / ``Embedding`` / ``LayerNorm`` / ``BatchNorm1d`` modules, so ``parameters()``,
    ``state_dict()``, ``apply()``, ``modules()``, hooks, ``model(...)``, ``torch.save(model)`` and ``.to()`` are torch's
    own.  The children only HOLD the parameters: ``forward`` hands them to the device kernels.

    Parameters are positional-compatible with the reference
    (``_gnn.py:453-464``); ``device`` (HIP ordinal) and ``max_chunk_structures`` are
    additions.  ``ref_structure`` only needs ``lattice``, ``positions``,
    ``atomic_numbers`` and ``num_atoms``.
    """

    # pylint: disable=too-many-arguments,too-many-positional-arguments,too-many-locals
    def __init__(self, ref_structure, cutoff: float, size_node_embedding: int,
                 size_edge_embedding: int, num_message_passes: int,
                 gaussian_filter_start: float, gaussian_filter_end: float,
                 mean_polarizability: NDArray[np.float64],
                 stddev_polarizability: NDArray[np.float64], device: int | None = None,
                 max_chunk_structures: int = 0):
        if cutoff <= 0:
            raise ValueError(f"invalid cutoff: {cutoff} <= 0")
        if size_node_embedding <= 0:
            raise ValueError(f"invalid size_node_embedding: {size_node_embedding} <= 0")
        if size_edge_embedding <= 0:
            raise ValueError(f"invalid size_edge_embedding: {size_edge_embedding} <= 0")
        if num_message_passes <= 0:
            raise ValueError(f"invalid num_message_passes: {num_message_passes} <= 0")
        if gaussian_filter_start < 0:
            raise ValueError(f"invalid gaussian_filter_start: {gaussian_filter_start} < 0")
        if gaussian_filter_end <= gaussian_filter_start:
            raise ValueError(
                f"invalid gaussian_filter_end: {gaussian_filter_end} <= gaussian_filter_start")
        verify_ndarray_shape("mean_polarizability", mean_polarizability, (3, 3))
        verify_ndarray_shape("stddev_polarizability", stddev_polarizability, (3, 3))
        _lib.load()  # fail loudly before any work if the HIP library is absent
        torch.nn.Module.__init__(self)

        self._ref_structure = ref_structure
        self._cutoff = cutoff
        self._fn = int(size_node_embedding)
        self._fe = int(size_edge_embedding)
        self._passes = int(num_message_passes)
        self._mean_polarizability = np.array(mean_polarizability, dtype=np.float64)
        self._stddev_polarizability = np.array(stddev_polarizability, dtype=np.float64)
        self._device = device
        self._max_chunk = int(max_chunk_structures)

        # frozen graph of the reference structure (_gnn.py:489-499)
        # (pair test on the device when there is one; the numpy restatement of the same float32
        # arithmetic serves GPU-less hosts, e.g. unit tests -- both are pinned by the fixtures)
        if torch.cuda.is_available():
            dev = device if device is not None else torch.cuda.current_device()
            edges = graph_utils.radius_graph_pbc_device(ref_structure.lattice, ref_structure.positions,
                                                        cutoff, dev)
        else:
            edges = graph_utils.radius_graph_pbc(ref_structure.lattice, ref_structure.positions, cutoff)
        self._ref_edge_indexes = np.vstack([np.zeros((1, edges.shape[1]), dtype=np.int64), edges])
        self._atom_type_map = graph_utils.atom_type_map(ref_structure.atomic_numbers)
        self._num_atom_types = int((self._atom_type_map >= 0).sum())

        self._gaussian_filter = (float(gaussian_filter_start), float(gaussian_filter_end))
        if self._fe < 2:
            raise ValueError("invalid size_edge_embedding: the Gaussian filter needs >= 2 steps")
        # the reference's module tree, created in its order so that the same torch.manual_seed gives the same initial
        # weights (_gnn.py:508-539): trainable entries are Parameters (torch.optim works on them), the Gaussian offsets
        # and the BatchNorm running statistics are buffers
        fn, fe, k = self._fn, self._fe, self._num_atom_types
        self._node_embedding = torch.nn.Sequential(torch.nn.Embedding(k, fn), _Activation(), torch.nn.Linear(fn, fn),
                                                   _Activation(), torch.nn.Linear(fn, fn))
        self._edge_embedding = _GaussianFilter(*self._gaussian_filter, fe)
        self._node_blocks = torch.nn.ModuleList(_NodeBlockParams(fn, fe) for _ in range(self._passes))
        self._edge_blocks = torch.nn.ModuleList(_EdgeBlockParams(fn, fe) for _ in range(self._passes))
        self._to_polarizability_embedding = torch.nn.Sequential(
            torch.nn.Linear(fe, fe), torch.nn.BatchNorm1d(fe), _Activation(), torch.nn.Linear(fe, fe), _Activation(),
            torch.nn.Linear(fe, 12))
        offset = self._edge_embedding.offset
        self._gauss_coefficient = -0.5 / (float(offset[1]) - float(offset[0])) ** 2  # _gnn.py:64
        self._device_ahead = False     # DeviceAdam stepped: the device weights are newer than the host tensors
        self._device_training = False  # gradients / optimiser state / BatchNorm buffers live in HBM
        self._device_anchor = torch.zeros(1, requires_grad=True)
        self._device_batches_tracked = 0
        self._handle = None
        self._profiling = 0
        self._uploaded_version = None
        # (a fresh torch Module is in training mode; calc_polarizabilities switches to eval)

    @property
    def _state_store(self) -> "OrderedDict[str, torch.Tensor]":
        """The module's own parameters and buffers, by ``state_dict`` key and in its order (the tensors themselves)."""
        return torch.nn.Module.state_dict(self, keep_vars=True)

    @property
    def _state(self) -> "OrderedDict[str, torch.Tensor]":
        """The module's parameters and buffers by ``state_dict`` key (the tensors themselves), refreshed from the device
        first when a device-resident optimiser has moved the weights (``DeviceAdam``)."""
        if self._device_ahead:
            self._sync_from_device()
        return self._state_store

    def _sync_from_device(self) -> None:
        store = self._state_store
        blob = np.empty(sum(v.numel() for v in store.values() if v.is_floating_point()), dtype=np.float32)
        rc = _lib.load().rn_potgnn_get_weights(self._handle, _ptr(blob), blob.size)
        _lib.check(rc, self._handle, "rn_potgnn_get_weights")
        offset = 0
        with torch.no_grad():
            for value in store.values():
                if value.is_floating_point():
                    n = value.numel()
                    value.copy_(torch.from_numpy(blob[offset:offset + n].reshape(tuple(value.shape))).to(value.device))
                    offset += n
            store["_to_polarizability_embedding.1.num_batches_tracked"].fill_(self._device_batches_tracked)
        self._device_ahead = False
        self._uploaded_version = sum(int(v._version) for v in store.values())  # device == host again

    def reset_parameters(self) -> None:
        """Re-initialise every parameter and buffer (``_gnn.py:559-566``: ``reset`` on every child, in order)."""
        if self._device_ahead:
            self._sync_from_device()
        for module in self.modules():
            if module is not self and hasattr(module, "reset_parameters"):
                module.reset_parameters()
        with torch.no_grad():
            self._edge_embedding.offset.copy_(torch.linspace(*self._gaussian_filter, self._fe, dtype=torch.float32))

    # ------------------------------------------------------------------ properties
    @property
    def ref_edge_indexes(self) -> np.ndarray:
        """``int64[3,E]`` (graph, a, b) as ``PotGNN._ref_edge_indexes`` (_gnn.py:492)."""
        return self._ref_edge_indexes.copy()

    @property
    def num_edges(self) -> int:
        return int(self._ref_edge_indexes.shape[1])

    @property
    def num_atoms(self) -> int:
        return int(self._ref_structure.num_atoms)

    @property
    def atom_type_map(self) -> np.ndarray:
        return self._atom_type_map.copy()

    @property
    def gauss_coefficient(self) -> float:
        return self._gauss_coefficient

    # ------------------------------------------------------------------ parameters
    _BUFFER_SUFFIXES = ("_edge_embedding.offset", "running_mean", "running_var", "num_batches_tracked")

    @classmethod
    def _is_trainable(cls, key: str) -> bool:
        return not key.endswith(cls._BUFFER_SUFFIXES)

    # torch.nn.Module's own parameters / named_parameters / state_dict / load_state_dict / apply / train / eval, with one
    # addition: after a device-resident optimiser step the host tensors are refreshed from the device first
    def _fresh(self) -> None:
        if self._device_ahead:
            self._sync_from_device()

    def parameters(self, recurse: bool = True):  # pylint: disable=missing-function-docstring
        self._fresh()
        return super().parameters(recurse)

    def named_parameters(self, *args, **kwargs):  # pylint: disable=missing-function-docstring
        self._fresh()
        return super().named_parameters(*args, **kwargs)

    def state_dict(self, *args, **kwargs):
        """Same keys, shapes and order as the reference's ``state_dict()``."""
        self._fresh()
        return super().state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """``torch.nn.Module.load_state_dict`` (numpy arrays are accepted as values too); the Parameter objects stay the
        same, so optimisers holding them stay valid."""
        self._fresh()
        state = OrderedDict((k, v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))) for k, v in state_dict.items())
        return super().load_state_dict(state, strict=strict, assign=assign)

    # ------------------------------------------------------------------ device handle
    def _weights_blob(self) -> np.ndarray:
        parts = [v.detach().cpu().numpy().astype(np.float32).ravel() for v in self._state.values()
                 if v.is_floating_point()]
        return np.ascontiguousarray(np.concatenate(parts))

    def _version(self) -> int:
        return sum(int(v._version) for v in self._state.values())  # bumped by in-place updates

    def _ensure_handle(self):
        if self._handle is not None:
            if self._device_ahead:  # the device copy is the current one (see the _state property)
                return self._handle
            if self._uploaded_version != self._version():  # e.g. after optimizer.step()
                blob = self._weights_blob()
                rc = _lib.load().rn_potgnn_set_weights(self._handle, _ptr(blob), blob.size)
                _lib.check(rc, self._handle, "rn_potgnn_set_weights")
                self._uploaded_version = self._version()
            return self._handle
        lib = _lib.load()
        device = self._device
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        cfg = _lib.Config(self.num_atoms, self.num_edges, self._num_atom_types, self._fn, self._fe,
                          self._passes, self._gauss_coefficient, self._max_chunk, int(device))
        blob = self._weights_blob()
        expected = lib.rn_potgnn_weight_count(C.byref(cfg))
        if blob.size != expected:
            raise RuntimeError(f"weight blob has {blob.size} floats, library expects {expected}")
        edge_a = np.ascontiguousarray(self._ref_edge_indexes[1], dtype=np.int32)
        edge_b = np.ascontiguousarray(self._ref_edge_indexes[2], dtype=np.int32)
        types = np.ascontiguousarray(
            self._atom_type_map[np.asarray(self._ref_structure.atomic_numbers)], dtype=np.int32)
        lattice = np.ascontiguousarray(self._ref_structure.lattice, dtype=np.float64)
        mean = np.ascontiguousarray(self._mean_polarizability)
        std = np.ascontiguousarray(self._stddev_polarizability)
        handle = C.c_void_p()
        rc = lib.rn_potgnn_create(C.byref(cfg), _ptr(edge_a), _ptr(edge_b), _ptr(types),
                                  _ptr(lattice), _ptr(blob), blob.size, _ptr(mean), _ptr(std),
                                  C.byref(handle))
        _lib.check(rc, None, "rn_potgnn_create")
        self._handle = handle
        self._uploaded_version = self._version()
        if self._profiling:
            lib.rn_potgnn_set_profiling(handle, self._profiling)
        return handle

    # -- copies and pickles carry the host state only (copy.deepcopy(model), torch.save(model)): the
    #    device handle, its optimiser state and the data-parallel hooks belong to this object
    def __getstate__(self):
        self._fresh()  # (the host tensors are fetched from the device first when that copy is ahead)
        state = dict(self.__dict__)
        for key in ("_handle", "_dp_group", "_dp_callback", "_dp_installed_on"):
            state.pop(key, None)
        state.update(_device_ahead=False, _device_training=False, _uploaded_version=None)
        return state

    def __setstate__(self, state):
        torch.nn.Module.__setstate__(self, state)
        self.__dict__["_handle"] = None

    def _release(self) -> None:
        if getattr(self, "_handle", None) is not None:
            _lib.load().rn_potgnn_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:  # pylint: disable=broad-except
            pass

    # ------------------------------------------------------------------ evaluation
    def _check_positions(self, positions_batch) -> np.ndarray:
        verify_ndarray_shape("positions_batch", positions_batch, (None, self.num_atoms, 3))
        return np.ascontiguousarray(positions_batch, dtype=np.float64)

    def calc_polarizabilities(self, positions_batch: NDArray[np.float64], dtype=None,
                              progress: bool = False) -> NDArray[np.float64]:
        """Polarizabilities ``(S,3,3)`` for fractional positions ``(S,N,3)``
        (``_gnn.py:667-721``): host arrays in, host arrays out.

        ``progress=True`` shows the reference's progress bar (``tqdm``, unit " configs", ``_gnn.py:692,714-717``): the
        batch is then evaluated in blocks through the pipelined entry so that the bar moves; off by default -- one call
        is a fraction of a second for 10 000 frames.

        ``dtype`` is the arithmetic the model is evaluated in.  ``None`` follows
        ``torch.get_default_dtype()`` as the reference does (``_gnn.py:705-710``): float32 unless the
        caller ran ``torch.set_default_dtype(torch.float64)``; ``torch.float64`` / ``numpy.float64``
        selects the kernels instantiated for ``double`` (``rn_potgnn_calc_polarizabilities_f64``)."""
        pos = self._check_positions(positions_batch)
        if self.training:  # as the reference does (_gnn.py:686); not walked again per call (the reference's unchanged
            self.eval()    # Phonons loop makes 2 M calls of one structure each, dynamics/_phonon.py:93-106)
        out = np.empty((pos.shape[0], 3, 3), dtype=np.float64)
        handle = self._ensure_handle()
        if progress and pos.shape[0] > 0:
            from tqdm import tqdm
            entry = (_lib.load().rn_potgnn_calc_polarizabilities_f64 if _wants_float64(dtype)
                     else _lib.load().rn_potgnn_calc_polarizabilities)
            block = max(1, min(pos.shape[0], 2000))
            with tqdm(total=pos.shape[0], unit=" configs") as progress_bar:
                for first in range(0, pos.shape[0], block):
                    n = min(block, pos.shape[0] - first)
                    rc = entry(handle, _ptr(pos[first:first + n]), n, _ptr(out[first:first + n]))
                    _lib.check(rc, handle, "rn_potgnn_calc_polarizabilities")
                    progress_bar.update(n)
            return out
        if _wants_float64(dtype):
            rc = _lib.load().rn_potgnn_calc_polarizabilities_f64(handle, _ptr(pos), pos.shape[0], _ptr(out))
            _lib.check(rc, handle, "rn_potgnn_calc_polarizabilities_f64")
            return out
        rc = _lib.load().rn_potgnn_calc_polarizabilities(handle, _ptr(pos), pos.shape[0], _ptr(out))
        _lib.check(rc, handle, "rn_potgnn_calc_polarizabilities")
        return out

    @property
    def device_index(self) -> int:
        """HIP ordinal the model evaluates on."""
        if self._device is not None:
            return int(self._device)
        return torch.cuda.current_device() if torch.cuda.is_available() else 0

    def calc_polarizabilities_async(self, positions_batch: NDArray[np.float64],
                                    out: NDArray[np.float64]) -> None:
        """Pipelined form of ``calc_polarizabilities`` (``rn_potgnn_calc_polarizabilities_async``):
        enqueues copy-in, evaluation and copy-out and returns; ``out`` ``(S,3,3)`` is valid after
        ``wait()``.  ``positions_batch`` and ``out`` must be C-contiguous float64 arrays that stay
        alive and untouched until then (page-locked memory makes the copies asynchronous)."""
        verify_ndarray_shape("positions_batch", positions_batch, (None, self.num_atoms, 3))
        s = positions_batch.shape[0]
        for name, arr, shape in (("positions_batch", positions_batch, positions_batch.shape), ("out", out, (s, 3, 3))):
            if arr.dtype != np.float64 or not arr.flags.c_contiguous or tuple(arr.shape) != tuple(shape):
                raise ValueError(f"{name} must be a C-contiguous float64 array of shape {tuple(shape)}")
        self.eval()
        handle = self._ensure_handle()
        rc = _lib.load().rn_potgnn_calc_polarizabilities_async(handle, _ptr(positions_batch), s, _ptr(out))
        _lib.check(rc, handle, "rn_potgnn_calc_polarizabilities_async")

    def wait(self) -> None:
        """Block until every ``calc_polarizabilities_async`` call has delivered its result."""
        if self._handle is not None:
            rc = _lib.load().rn_potgnn_wait(self._handle)
            _lib.check(rc, self._handle, "rn_potgnn_wait")

    def calc_polarizabilities_to_device(self, positions_batch: NDArray[np.float64],
                                        out: torch.Tensor | None = None) -> torch.Tensor:
        """``calc_polarizabilities`` with the result left in HBM (``rn_potgnn_calc_polarizabilities_to_device``): host
        ``float64[S,N,3]`` in, device ``float64[S,3,3]`` out, float32 arithmetic.  The upload is pipelined with the
        kernels (positions cast to float32 while staged: bit-identical results, half the PCIe bytes); torch's current
        stream is ordered behind the evaluation, so the tensor can go straight into a collective."""
        pos = self._check_positions(positions_batch)
        self.eval()
        device = torch.device("cuda", self.device_index)
        if out is None:
            out = torch.empty((pos.shape[0], 3, 3), dtype=torch.float64, device=device)
        handle = self._ensure_handle()
        stream = torch.cuda.current_stream(device).cuda_stream
        rc = _lib.load().rn_potgnn_calc_polarizabilities_to_device(handle, _ptr(pos), pos.shape[0],
                                                                   C.c_void_p(out.data_ptr()), C.c_void_p(stream))
        _lib.check(rc, handle, "rn_potgnn_calc_polarizabilities_to_device")
        return out

    def calc_polarizabilities_device(self, positions: torch.Tensor, out: torch.Tensor | None = None,
                                     synchronize: bool = False, dtype=None) -> torch.Tensor:
        """Same computation on a device-resident ``float64[S,N,3]`` tensor; returns a device
        ``float64[S,3,3]`` tensor.  Work is enqueued on torch's current stream.  ``dtype`` is the
        arithmetic of the evaluation, resolved exactly as ``calc_polarizabilities`` does: ``None`` follows
        ``torch.get_default_dtype()`` (``_gnn.py:705-710``), ``torch.float64`` runs the kernels instantiated
        for ``double`` -- so the host, gloo and RCCL paths of ``ramannoodle_amd.parallel`` agree."""
        if not (positions.is_cuda and positions.dtype == torch.float64 and positions.is_contiguous()):
            raise ValueError("positions must be a contiguous float64 device tensor")
        if positions.dim() != 3 or tuple(positions.shape[1:]) != (self.num_atoms, 3):
            raise ValueError(f"positions_batch has wrong shape: {tuple(positions.shape)} != "
                             f"(_,{self.num_atoms},3)")
        s = positions.shape[0]
        if out is None:
            out = torch.empty((s, 3, 3), dtype=torch.float64, device=positions.device)
        handle = self._ensure_handle()
        stream = torch.cuda.current_stream(positions.device).cuda_stream
        if _wants_float64(dtype):
            rc = _lib.load().rn_potgnn_forward_device_f64(
                handle, C.c_void_p(positions.data_ptr()), s, C.c_void_p(out.data_ptr()), C.c_void_p(stream),
                int(synchronize))
            _lib.check(rc, handle, "rn_potgnn_forward_device_f64")
            return out
        rc = _lib.load().rn_potgnn_forward_device(
            handle, C.c_void_p(positions.data_ptr()), s, C.c_void_p(out.data_ptr()), None,
            C.c_void_p(stream), int(synchronize))
        _lib.check(rc, handle, "rn_potgnn_forward_device")
        return out

    def forward(self, lattice, atomic_numbers, positions) -> torch.Tensor:
        """Standardised 6-vectors ``[S,6]`` (``_gnn.py:617-665``).  ``lattice`` ``[S,3,3]`` is
        used per sample in the geometry, as the reference does (``_gnn.py:603-611``), and
        ``atomic_numbers`` ``[S,N]`` per sample for the node embedding
        (``_convert_to_atom_type``, ``_gnn.py:541-557,642-643``), in evaluation AND in training mode; the graph
        topology is the reference structure's.  An atomic number the model has no atom type for raises
        ``IndexError``, as the reference's ``Embedding`` does.

        Arithmetic: the reference's ``forward`` computes in the dtype of the model's parameters (``_gnn.py:493-494``):
        float32 unless the caller ran ``torch.set_default_dtype(torch.float64)`` before constructing the model.  So does this
        one in EVALUATION mode: a model whose parameters are float64 (built under that default, or ``model.double()``) runs
        the kernels instantiated for ``double`` (``rn_potgnn_forward_samples_f64``) and returns a float64 tensor; the device
        keeps float32 master weights, so float64 parameters take part rounded to float32 (exact for a state dict that was
        trained in float32).  In TRAINING mode ``forward`` evaluates in float32 whatever the parameters' dtype (the float64
        step is ``train_gradients_f64``).  ``calc_polarizabilities(..., dtype=...)`` follows the default dtype when
        ``dtype`` is ``None`` (``_gnn.py:705-710``), as does ``calc_polarizabilities_device``.
        An evaluation-mode ``forward`` (or any other evaluation) between a training-mode ``forward`` and its ``backward()``
        reuses the device workspace the pending step's tape refers to: run ``backward()`` first -- the backward of a step
        whose tape was overwritten raises ``ValueError`` ("needs a preceding train_forward").

        The result lives where the inputs live, as the reference's does (``_train.py:51-75`` moves a batch to the
        device and takes the loss there; ``test/tests/torch/test_gnn.py:130-160``): CUDA tensors are evaluated in
        place in HBM (``rn_potgnn_forward_samples_device``: no copy through the host) and a CUDA tensor comes back;
        host tensors / arrays give a host tensor."""
        pos_t, lat_t, zs_t = torch.as_tensor(positions), torch.as_tensor(lattice), torch.as_tensor(atomic_numbers)
        s = pos_t.shape[0] if pos_t.dim() == 3 else -1
        if pos_t.dim() != 3 or tuple(pos_t.shape[1:]) != (self.num_atoms, 3):
            verify_ndarray_shape("positions", pos_t.detach().cpu().numpy(), (None, self.num_atoms, 3))
        if tuple(lat_t.shape) != (s, 3, 3) or tuple(zs_t.shape) != (s, self.num_atoms):
            raise ValueError("lattice / atomic_numbers do not match positions")
        on_device = pos_t.is_cuda or lat_t.is_cuda or zs_t.is_cuda
        model_f64 = (not self.training) and next(self.parameters()).dtype == torch.float64
        if on_device and not self.training and not model_f64:
            return self._forward_on_device(lat_t, zs_t, pos_t)
        if on_device and self._device_training and getattr(self, "_dp_group", None) is None:
            # device-resident training on a device-resident batch: nothing crosses PCIe in either direction
            return _TrainStepOnDevice.apply(self, lat_t, zs_t, pos_t, self._device_anchor)
        out_device = next((t.device for t in (pos_t, lat_t, zs_t) if t.is_cuda), None)
        pos = np.ascontiguousarray(pos_t.detach().cpu().numpy(), dtype=np.float64)
        lat = np.ascontiguousarray(lat_t.detach().cpu().numpy(), dtype=np.float64)
        zs = zs_t.detach().cpu().numpy()
        same_species = (not s) or np.array_equal(zs, np.broadcast_to(
            np.asarray(self._ref_structure.atomic_numbers), zs.shape))
        same_lattice = (not s) or np.allclose(lat, self._ref_structure.lattice[None], rtol=1e-6, atol=1e-9)
        types = None
        if not same_species:
            zi = zs.astype(np.int64)
            if zi.min() < -self._atom_type_map.size or zi.max() >= self._atom_type_map.size:
                raise IndexError("atomic number outside the atom type map")
            types = np.ascontiguousarray(self._atom_type_map[zi], dtype=np.int32)
            if types.min() < 0:
                raise IndexError("index out of range in self: atomic_numbers holds a species the "
                                 "model has no atom type for")
        if self.training:
            extra = (None if same_lattice else lat, types)
            if self._device_training:  # gradients stay in HBM: nothing for autograd to route
                out = _TrainStep.apply(self, pos, extra, self._device_anchor)
            else:
                out = _TrainStep.apply(self, pos, extra, *self.parameters())
            return out.to(out_device) if out_device is not None else out
        handle = self._ensure_handle()
        if model_f64:  # the model's parameters are float64: evaluate in double, as the reference does
            out64 = np.empty((s, 6), dtype=np.float64)
            rc = _lib.load().rn_potgnn_forward_samples_f64(
                handle, None if same_lattice else _ptr(lat), None if types is None else _ptr(types), _ptr(pos), s, _ptr(out64))
            _lib.check(rc, handle, "rn_potgnn_forward_samples_f64")
            result = torch.from_numpy(out64)
            return result.to(out_device) if out_device is not None else result
        out = np.empty((s, 6), dtype=np.float32)
        if same_lattice and same_species:
            rc = _lib.load().rn_potgnn_forward(handle, _ptr(pos), s, _ptr(out))
            _lib.check(rc, handle, "rn_potgnn_forward")
        else:
            rc = _lib.load().rn_potgnn_forward_samples(
                handle, None if same_lattice else _ptr(lat), None if types is None else _ptr(types),
                _ptr(pos), s, _ptr(out))
            _lib.check(rc, handle, "rn_potgnn_forward_samples")
        return torch.from_numpy(out)

    def _device_inputs(self, lat_t: torch.Tensor, zs_t: torch.Tensor, pos_t: torch.Tensor):
        """What the device entries take, from tensors of which at least one is on a GPU: every check and conversion is a
        torch operation on the device.  Returns ``(device, S, positions f64 [S,N,3], lattices f32 [S,9] or None, atom types
        int32 [S,N] or None)``."""
        device = next(t.device for t in (pos_t, lat_t, zs_t) if t.is_cuda)
        if device.index is not None and self._device is not None and device.index != int(self._device):
            raise ValueError(f"inputs live on {device}, the model evaluates on cuda:{int(self._device)}")
        s = pos_t.shape[0]
        pos = pos_t.detach().to(device=device, dtype=torch.float64).contiguous()
        lat = lat_t.detach().to(device=device, dtype=torch.float64)
        zs = zs_t.detach().to(device=device)
        if s == 0:
            return device, s, pos, None, None
        ref_lat = torch.as_tensor(np.asarray(self._ref_structure.lattice, dtype=np.float64), device=device)
        ref_zs = torch.as_tensor(np.asarray(self._ref_structure.atomic_numbers), device=device).to(zs.dtype)
        same_lattice = bool(torch.allclose(lat, ref_lat.expand_as(lat), rtol=1e-6, atol=1e-9))
        same_species = bool(torch.equal(zs, ref_zs.expand_as(zs)))
        lat32 = types = None
        if not same_lattice:
            lat32 = lat.to(torch.float32).reshape(s, 9).contiguous()  # the reference's forward computes in float32
        if not same_species:
            zi = zs.to(torch.int64)
            size = self._atom_type_map.size
            if int(zi.min()) < -size or int(zi.max()) >= size:
                raise IndexError("atomic number outside the atom type map")
            tmap = torch.as_tensor(np.asarray(self._atom_type_map, dtype=np.int32), device=device)
            types = tmap[zi].contiguous()  # (negative atomic numbers index from the end, as numpy does on the host path)
            if int(types.min()) < 0:
                raise IndexError("index out of range in self: atomic_numbers holds a species the "
                                 "model has no atom type for")
        return device, s, pos, lat32, types

    def _forward_on_device(self, lat_t: torch.Tensor, zs_t: torch.Tensor, pos_t: torch.Tensor) -> torch.Tensor:
        """Evaluation-mode ``forward`` on device-resident inputs: the kernels read the tensors where they are, the ``[S,6]``
        result is a device tensor."""
        device, s, pos, lat32, types = self._device_inputs(lat_t, zs_t, pos_t)
        out = torch.empty((s, 6), dtype=torch.float32, device=device)
        if s == 0:
            return out
        handle = self._ensure_handle()
        stream = torch.cuda.current_stream(device).cuda_stream
        rc = _lib.load().rn_potgnn_forward_samples_device(
            handle, None if lat32 is None else C.c_void_p(lat32.data_ptr()),
            None if types is None else C.c_void_p(types.data_ptr()), C.c_void_p(pos.data_ptr()), s,
            C.c_void_p(out.data_ptr()), C.c_void_p(stream), 1)
        _lib.check(rc, handle, "rn_potgnn_forward_samples_device")
        return out

    def _train_forward_on_device(self, lat_t: torch.Tensor, zs_t: torch.Tensor, pos_t: torch.Tensor) -> torch.Tensor:
        """Training-mode ``forward`` of a device-resident batch with device-resident weights
        (``rn_potgnn_train_forward_samples_device``): enqueued behind torch's current stream, no synchronisation."""
        device, s, pos, lat32, types = self._device_inputs(lat_t, zs_t, pos_t)
        if s == 0:
            raise ValueError("a training batch needs at least one structure")
        out = torch.empty((s, 6), dtype=torch.float32, device=device)
        handle = self._ensure_handle()
        self._install_reducer(handle)  # (no data-parallel group here: clears one that was installed before)
        stream = torch.cuda.current_stream(device).cuda_stream
        rc = _lib.load().rn_potgnn_train_forward_samples_device(
            handle, None if lat32 is None else C.c_void_p(lat32.data_ptr()),
            None if types is None else C.c_void_p(types.data_ptr()), C.c_void_p(pos.data_ptr()), s,
            C.c_void_p(out.data_ptr()), C.c_void_p(stream))
        _lib.check(rc, handle, "rn_potgnn_train_forward_samples_device")
        self._device_batches_tracked += 1  # (the library updated the running statistics where they live)
        self._device_ahead = True
        return out

    def _train_backward_on_device(self, grad_out: torch.Tensor) -> None:
        dvec6 = grad_out.detach().to(dtype=torch.float32).contiguous()
        stream = torch.cuda.current_stream(dvec6.device).cuda_stream
        rc = _lib.load().rn_potgnn_train_backward_samples_device(self._handle, C.c_void_p(dvec6.data_ptr()),
                                                                 C.c_void_p(stream))
        _lib.check(rc, self._handle, "rn_potgnn_train_backward_samples_device")

    # -- training step pieces used by _TrainStep ------------------------------------------
    def enable_data_parallel(self, group=None) -> None:
        """Data-parallel training over the ranks of ``group`` (default process group): the
        readout BatchNorm then uses the statistics of the global batch (all-reduced column
        sums inside the device step, ``rn_potgnn_set_stat_reducer``); gradient averaging is
        ``ramannoodle_amd.parallel.average_gradients`` (``train_single_epoch`` calls it).
        ``None`` world / single rank: no effect."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            self._dp_group, self._dp_callback = None, None
        else:
            on_gpu = dist.get_backend(group) == "nccl"

            def _sum_over_ranks(values, count, _ctx):
                try:
                    view = np.ctypeslib.as_array(values, shape=(count,))
                    tensor = torch.from_numpy(view.copy())
                    if on_gpu:
                        tensor = tensor.cuda()
                    dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
                    view[:] = tensor.cpu().numpy()
                    return 0
                except Exception:  # pylint: disable=broad-except  (must not unwind through C)
                    return 1

            self._dp_group = group if group is not None else dist.group.WORLD
            self._dp_callback = _REDUCE_FN(_sum_over_ranks)  # keep the thunk alive
        self._dp_installed_on = None

    @property
    def data_parallel_group(self):
        """Process group set by ``enable_data_parallel`` (``None`` = single process)."""
        return getattr(self, "_dp_group", None)

    def _install_reducer(self, handle) -> None:
        callback = getattr(self, "_dp_callback", None)
        if getattr(self, "_dp_installed_on", None) == (handle, id(callback)):
            return
        rc = _lib.load().rn_potgnn_set_stat_reducer(handle, C.cast(callback, C.c_void_p) if callback else None,
                                                    None)
        _lib.check(rc, handle, "rn_potgnn_set_stat_reducer")
        self._dp_installed_on = (handle, id(callback))

    def _train_forward(self, pos: np.ndarray, lat: np.ndarray | None = None, types: np.ndarray | None = None) -> np.ndarray:
        handle = self._ensure_handle()
        self._install_reducer(handle)
        out = np.empty((pos.shape[0], 6), dtype=np.float32)
        mean = np.empty(self._fe, dtype=np.float32)
        var = np.empty(self._fe, dtype=np.float32)
        if lat is None and types is None:
            rc = _lib.load().rn_potgnn_train_forward(handle, _ptr(pos), pos.shape[0], _ptr(out),
                                                     _ptr(mean), _ptr(var))
            _lib.check(rc, handle, "rn_potgnn_train_forward")
        else:  # a lattice and / or atom types per sample (_gnn.py:603-611, 541-557)
            rc = _lib.load().rn_potgnn_train_forward_samples(
                handle, None if lat is None else _ptr(lat), None if types is None else _ptr(types), _ptr(pos),
                pos.shape[0], _ptr(out), _ptr(mean), _ptr(var))
            _lib.check(rc, handle, "rn_potgnn_train_forward_samples")
        if self._device_training:  # the library updated the running statistics where they live
            self._device_batches_tracked += 1
            self._device_ahead = True
            return out
        # BatchNorm1d running statistics, torch semantics (momentum 0.1, unbiased variance)
        rows = int(round(_lib.load().rn_potgnn_train_row_count(handle)))  # all ranks' rows
        pre = "_to_polarizability_embedding.1."
        with torch.no_grad():
            running_mean, running_var = self._state[pre + "running_mean"], self._state[pre + "running_var"]
            running_mean.mul_(0.9).add_(torch.from_numpy(mean).to(running_mean.device) * 0.1)
            unbiased = torch.from_numpy(var).to(running_var.device) * (rows / max(rows - 1, 1))
            running_var.mul_(0.9).add_(unbiased * 0.1)
            self._state[pre + "num_batches_tracked"].add_(1)
        self._uploaded_version = None  # buffers changed: re-upload before the next evaluation
        return out

    def _train_backward(self, dvec6: np.ndarray):
        handle = self._handle
        lib = _lib.load()
        blob = np.empty(sum(v.numel() for v in self._state.values() if v.is_floating_point()),
                        dtype=np.float32)
        rc = lib.rn_potgnn_train_backward(handle, _ptr(dvec6), _ptr(blob))
        _lib.check(rc, handle, "rn_potgnn_train_backward")
        grads, offset = [], 0
        for key, value in self._state.items():
            if not value.is_floating_point():
                continue
            n = value.numel()
            if self._is_trainable(key):
                grads.append(torch.from_numpy(blob[offset:offset + n].reshape(tuple(value.shape)).copy()))
            offset += n
        return grads

    # -- device-resident optimisation (DeviceAdam) ------------------------------------------
    def enable_device_training(self, enabled: bool = True) -> None:
        """Keep parameter gradients, optimiser state and BatchNorm running statistics in HBM
        (``rn_potgnn_train_backward_device`` / ``rn_potgnn_adam_step``): a training step then
        moves only positions, outputs and ``dL/dout`` across PCIe.  ``parameters()`` /
        ``state_dict()`` fetch the current values on demand; ``p.grad`` is not populated."""
        handle = self._ensure_handle()
        if not enabled:
            _ = self._state  # bring the host copy up to date before handing control back
        rc = _lib.load().rn_potgnn_set_device_training(handle, int(bool(enabled)))
        _lib.check(rc, handle, "rn_potgnn_set_device_training")
        self._device_training = bool(enabled)
        self._device_batches_tracked = int(
            self._state_store["_to_polarizability_embedding.1.num_batches_tracked"])

    def _train_backward_device(self, dvec6: np.ndarray) -> None:
        rc = _lib.load().rn_potgnn_train_backward_device(self._handle, _ptr(dvec6))
        _lib.check(rc, self._handle, "rn_potgnn_train_backward_device")

    def device_gradients(self) -> torch.Tensor:
        """The float32 gradient buffer of the last device backward, as a CUDA tensor that
        aliases the library's memory (packed layout; for in-place all-reduce over RCCL)."""
        ptr, count = C.c_void_p(), C.c_size_t()
        rc = _lib.load().rn_potgnn_gradient_buffer(self._handle, C.byref(ptr), C.byref(count))
        _lib.check(rc, self._handle, "rn_potgnn_gradient_buffer")
        return _alias_device_buffer(int(ptr.value), int(count.value), self.device_index)

    def _adam_step(self, lr, beta1, beta2, eps, weight_decay, step) -> None:
        rc = _lib.load().rn_potgnn_adam_step(self._handle, lr, beta1, beta2, eps, weight_decay, step)
        _lib.check(rc, self._handle, "rn_potgnn_adam_step")
        self._device_ahead = True

    def train_gradients_f64(self, positions, targets, lattice=None, atomic_numbers=None):
        """One training step's forward and backward in float64 on the device
        (``rn_potgnn_train_forward_f64`` / ``_backward_f64``) for an MSE loss against ``targets``
        ``[S,6]``: returns ``(out [S,6] float64, loss, {parameter name: gradient float64})``.
        ``lattice`` ``[S,3,3]`` / ``atomic_numbers`` ``[S,N]``: per-sample inputs as ``forward`` takes them.
        Validation aid: what the float32 step is measured against; running statistics are not
        updated."""
        pos = np.ascontiguousarray(positions, dtype=np.float64)
        verify_ndarray_shape("positions", pos, (None, self.num_atoms, 3))
        tgt = np.ascontiguousarray(targets, dtype=np.float64)
        s = pos.shape[0]
        handle = self._ensure_handle()
        self._install_reducer(handle)
        lib = _lib.load()
        out = np.empty((s, 6), dtype=np.float64)
        mean, var = np.empty(self._fe), np.empty(self._fe)
        lat = None if lattice is None else np.ascontiguousarray(lattice, dtype=np.float64).reshape(s, 3, 3)
        types = None
        if atomic_numbers is not None:
            types = np.ascontiguousarray(self._atom_type_map[np.asarray(atomic_numbers, dtype=np.int64)], dtype=np.int32)
        if lat is None and types is None:
            rc = lib.rn_potgnn_train_forward_f64(handle, _ptr(pos), s, _ptr(out), _ptr(mean), _ptr(var))
            _lib.check(rc, handle, "rn_potgnn_train_forward_f64")
        else:
            rc = lib.rn_potgnn_train_forward_samples_f64(handle, None if lat is None else _ptr(lat),
                                                         None if types is None else _ptr(types), _ptr(pos), s, _ptr(out),
                                                         _ptr(mean), _ptr(var))
            _lib.check(rc, handle, "rn_potgnn_train_forward_samples_f64")
        loss = float(np.mean((out - tgt) ** 2))
        dvec6 = np.ascontiguousarray(2.0 * (out - tgt) / out.size)
        blob = np.empty(sum(v.numel() for v in self._state.values() if v.is_floating_point()), dtype=np.float64)
        rc = lib.rn_potgnn_train_backward_f64(handle, _ptr(dvec6), _ptr(blob))
        _lib.check(rc, handle, "rn_potgnn_train_backward_f64")
        grads, offset = {}, 0
        for key, value in self._state.items():
            if not value.is_floating_point():
                continue
            n = value.numel()
            if self._is_trainable(key):
                grads[key] = blob[offset:offset + n].reshape(tuple(value.shape)).copy()
            offset += n
        return out, loss, grads

    def calc_raman_tensors(self, ref_positions, displacements,
                           delta: float = RAMAN_TENSOR_CENTRAL_DIFFERENCE,
                           method: str = "finite-difference") -> NDArray[np.float64]:
        """Raman tensors of all modes in one device call.

        ``method="finite-difference"`` (default, the reference's definition):
        ``(alpha(r + delta d_m) - alpha(r - delta d_m)) / delta`` in one float64 batch of the
        ``2M`` displaced cells (``dynamics/_phonon.py:93-106``).
        ``method="analytic"``: ``2 (d alpha / d r) . d_m`` from one forward and one reverse
        pass (equal up to ``O(delta^2)``).
        """
        self._check_raman_arguments(ref_positions, displacements, delta=delta, method=method)
        ref = np.ascontiguousarray(ref_positions, dtype=np.float64)
        disp = np.ascontiguousarray(displacements, dtype=np.float64)
        out = np.empty((disp.shape[0], 3, 3), dtype=np.float64)
        handle = self._ensure_handle()
        lib = _lib.load()
        if method == "analytic":
            rc = lib.rn_potgnn_raman_tensors_analytic(handle, _ptr(ref), _ptr(disp), disp.shape[0],
                                                      _ptr(out))
            _lib.check(rc, handle, "rn_potgnn_raman_tensors_analytic")
        elif method == "finite-difference":
            rc = lib.rn_potgnn_raman_tensors(handle, _ptr(ref), _ptr(disp), disp.shape[0],
                                             float(delta), _ptr(out))
            _lib.check(rc, handle, "rn_potgnn_raman_tensors")
        else:
            raise ValueError(f"unsupported method: {method}")
        return out

    def _check_raman_arguments(self, ref_positions, displacements, delta: float = RAMAN_TENSOR_CENTRAL_DIFFERENCE,
                               method: str = "finite-difference", **unknown) -> None:
        """Argument checks of ``calc_raman_tensors`` (also run by ``parallel.calc_raman_tensors_sharded`` on every
        rank before it branches, so that all ranks raise together)."""
        if unknown:
            raise TypeError(f"calc_raman_tensors() got unexpected keyword arguments: {sorted(unknown)}")
        verify_ndarray_shape("ref_positions", ref_positions, (self.num_atoms, 3))
        verify_ndarray_shape("displacements", displacements, (None, self.num_atoms, 3))
        if method not in ("finite-difference", "analytic"):
            raise ValueError(f"unsupported method: {method}")
        if method == "finite-difference" and not (np.isfinite(delta) and float(delta) != 0.0):
            raise ValueError(f"delta must be finite and non-zero, not {delta}")

    def alpha_jacobian(self, positions, float64: bool = True) -> NDArray[np.float64]:
        """``d vec6_k / d x_{n,c}`` of the standardised 6-vector at one structure,
        ``(6, N, 3)``, by reverse-mode differentiation on the device."""
        verify_ndarray_shape("positions", positions, (self.num_atoms, 3))
        pos = np.ascontiguousarray(positions, dtype=np.float64)
        out = np.empty((6, self.num_atoms, 3), dtype=np.float64)
        handle = self._ensure_handle()
        rc = _lib.load().rn_potgnn_alpha_jacobian(handle, _ptr(pos), int(float64), _ptr(out))
        _lib.check(rc, handle, "rn_potgnn_alpha_jacobian")
        return out

    # ------------------------------------------------------------------ introspection
    @property
    def num_triplets(self) -> int:
        """Number of edge triplets (k->j, j->i, i != k) of the frozen graph."""
        return int(_lib.load().rn_potgnn_num_triplets(self._ensure_handle()))

    def triplets(self):
        """Edge triplets as the device enumerates them, re-sorted into the reference's
        order: ``(i, j, idx_i, idx_j, idx_k, slot5, slot6)`` int64 arrays with the meaning of
        ``BatchTriplets._ref_triplets`` (``_utils.py:161-168``)."""
        handle = self._ensure_handle()
        lib = _lib.load()
        t = int(lib.rn_potgnn_num_triplets(handle))
        arrs = [np.empty(t, dtype=np.int32) for _ in range(5)]
        rc = lib.rn_potgnn_debug_triplets(handle, *[_ptr(a) for a in arrs])
        _lib.check(rc, handle, "rn_potgnn_debug_triplets")
        idx_i, idx_j, idx_k, slot5, slot6 = [a.astype(np.int64) for a in arrs]
        order = np.lexsort((idx_k, slot6))  # by source edge (j->i), then k
        return (self._ref_edge_indexes[2].copy(), self._ref_edge_indexes[1].copy(),
                idx_i[order], idx_j[order], idx_k[order], slot5[order], slot6[order])

    def device_triplets_raw(self):
        """Triplets in device aggregation order (grouped by destination edge)."""
        handle = self._ensure_handle()
        lib = _lib.load()
        t = int(lib.rn_potgnn_num_triplets(handle))
        arrs = [np.empty(t, dtype=np.int32) for _ in range(5)]
        _lib.check(lib.rn_potgnn_debug_triplets(handle, *[_ptr(a) for a in arrs]), handle,
                   "rn_potgnn_debug_triplets")
        return arrs

    def debug_stage(self, stage: int, index: int = 0) -> np.ndarray:
        """Intermediate of the last evaluation's last chunk (see ``rn_potgnn.h``)."""
        handle = self._ensure_handle()
        lib = _lib.load()
        cap = 64 * 1024 * 1024
        buf = np.empty(cap, dtype=np.float32)
        rows, cols = C.c_int64(), C.c_int64()
        rc = lib.rn_potgnn_debug_stage(handle, stage, index, _ptr(buf), cap, C.byref(rows),
                                       C.byref(cols))
        _lib.check(rc, handle, "rn_potgnn_debug_stage")
        return buf[: rows.value * cols.value].reshape(rows.value, cols.value).copy()

    def config_flags(self) -> dict:
        """Which kernel variants the handle selected (``rn_potgnn_config_flags``)."""
        flags = _lib.load().rn_potgnn_config_flags(self._ensure_handle())
        return {"fused_edge_block": bool(flags & 1), "folded_gate_scale": bool(flags & 2),
                "split_f16_mfma": bool(flags & 4), "narrow_kernels": bool(flags & 8),
                "mfma_range_fallback": bool(flags & 16), "pipelined_edge_block": bool(flags & 32),
                "twelve_wave_edge_block": bool(flags & 64), "experiment_kernels": bool(flags & 128),
                "role_split_edge_block": bool(flags & 256), "atom_owning_node_block": bool(flags & 512),
                "split_f16_pair_rows": bool(flags & 1024)}

    def set_profiling(self, mode: int) -> None:
        """0 = off, 1 = HIP-event timing of every kernel launch, 100+k = kernel k only."""
        self._profiling = int(mode)
        if self._handle is not None:
            _lib.load().rn_potgnn_set_profiling(self._handle, self._profiling)

    def kernel_times(self) -> dict:
        """``{kernel: (milliseconds, launches)}`` accumulated since ``set_profiling``."""
        handle = self._ensure_handle()
        lib = _lib.load()
        names = (C.c_char_p * 16)()
        ms = (C.c_double * 16)()
        launches = (C.c_int64 * 16)()
        n = lib.rn_potgnn_kernel_times(handle, names, ms, launches, 16)
        return {names[i].decode(): (ms[i], launches[i]) for i in range(max(n, 0))}


class _TrainStep(torch.autograd.Function):
    """``PotGNN.forward`` in training mode: forward and backward both run on the device
    (``rn_potgnn_train_forward`` / ``rn_potgnn_train_backward``); autograd only routes the
    parameter gradients."""

    @staticmethod
    def forward(ctx, model, pos, extra, *params):  # pylint: disable=arguments-differ
        ctx.model = model
        return torch.from_numpy(model._train_forward(pos, *extra))

    @staticmethod
    def backward(ctx, grad_out):  # pylint: disable=arguments-differ
        dvec6 = np.ascontiguousarray(grad_out.detach().cpu().numpy(), dtype=np.float32)
        if ctx.model._device_training:
            ctx.model._train_backward_device(dvec6)
            return (None, None, None, None)
        grads = ctx.model._train_backward(dvec6)
        params = [p for _, p in torch.nn.Module.named_parameters(ctx.model)]
        return (None, None, None, *[g.to(device=p.device, dtype=p.dtype) for g, p in zip(grads, params)])


class _TrainStepOnDevice(torch.autograd.Function):
    """``PotGNN.forward`` in training mode on CUDA tensors with ``DeviceAdam``: the batch, the result, the cotangents and
    the parameter gradients all stay in HBM (``rn_potgnn_train_forward_samples_device`` /
    ``rn_potgnn_train_backward_samples_device``); ``anchor`` only gives autograd a reason to call ``backward``."""

    @staticmethod
    def forward(ctx, model, lat_t, zs_t, pos_t, anchor):  # pylint: disable=arguments-differ,unused-argument
        ctx.model = model
        return model._train_forward_on_device(lat_t, zs_t, pos_t)

    @staticmethod
    def backward(ctx, grad_out):  # pylint: disable=arguments-differ
        ctx.model._train_backward_on_device(grad_out)
        return (None, None, None, None, None)


class _DeviceSpan:  # pylint: disable=too-few-public-methods
    """``__cuda_array_interface__`` carrier for a float32 range of library-owned device memory."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False),
                                         "version": 3, "strides": None}


def _alias_device_buffer(ptr: int, count: int, device_index: int) -> torch.Tensor:
    return torch.as_tensor(_DeviceSpan(ptr, count), device=torch.device("cuda", device_index))


class DeviceAdam:
    """``torch.optim.Adam`` (no amsgrad) applied where the weights live.  Constructing it switches
    the model to device-resident training; ``step()`` is one ``rn_potgnn_adam_step`` (Adam on the
    packed weights, refresh of everything derived from them), ``zero_grad()`` is a no-op (each
    backward overwrites the gradient buffer).  Drop-in for the optimiser argument of
    ``train_single_epoch`` (``pmodel/torch/_train.py:63-76``)."""

    def __init__(self, model: PotGNN, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameters")
        self.model = model
        self.param_groups = [dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps),
                                  weight_decay=float(weight_decay))]
        self.steps = 0
        model.enable_device_training(True)

    def step(self) -> None:
        group = self.param_groups[0]
        self.steps += 1
        self.model._adam_step(group["lr"], group["betas"][0], group["betas"][1], group["eps"],
                              group["weight_decay"], self.steps)

    def zero_grad(self, set_to_none: bool = True) -> None:  # pylint: disable=unused-argument
        return None
