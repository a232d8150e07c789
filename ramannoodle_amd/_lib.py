"""ctypes binding of ``librn_potgnn.so`` (the C ABI declared in ``include/rn_potgnn.h``).

There is no CPU fallback: if the library is missing or a device call fails the caller
gets a ``DeviceError``.
"""
from __future__ import annotations

import ctypes as C
import os

from ramannoodle_amd.exceptions import DeviceError

_LIB_PATH = os.environ.get(
    "RN_POTGNN_LIB", os.path.join(os.path.dirname(os.path.abspath(__file__)), "librn_potgnn.so"))

RN_OK = 0
RN_ERR_INVALID_ARGUMENT = -1
RN_ERR_UNSUPPORTED = -2
RN_ERR_NO_DEVICE = -3
RN_ERR_HIP = -4
RN_ERR_OUT_OF_MEMORY = -5


class Config(C.Structure):
    """``rn_potgnn_config``."""

    _fields_ = [
        ("num_atoms", C.c_int32),
        ("num_edges", C.c_int32),
        ("num_atom_types", C.c_int32),
        ("size_node_embedding", C.c_int32),
        ("size_edge_embedding", C.c_int32),
        ("num_message_passes", C.c_int32),
        ("gauss_coefficient", C.c_double),
        ("max_chunk_structures", C.c_int32),
        ("device", C.c_int32),
    ]


# symbol -> (restype, argtypes); must list every function declared in include/*.h
_P = C.c_void_p
SIGNATURES = {
    "rn_potgnn_radius_graph": (C.c_int, [_P, _P, C.c_int32, C.c_double, C.c_int, _P]),
    "rn_potgnn_config_flags": (C.c_int, [_P]),
    "rn_potgnn_set_stat_reducer": (C.c_int, [_P, _P, _P]),
    "rn_potgnn_train_row_count": (C.c_double, [_P]),
    "rn_potgnn_weight_count": (C.c_size_t, [C.POINTER(Config)]),
    "rn_potgnn_create": (C.c_int, [C.POINTER(Config), _P, _P, _P, _P, _P, C.c_size_t, _P, _P,
                                   C.POINTER(_P)]),
    "rn_potgnn_destroy": (None, [_P]),
    "rn_potgnn_calc_polarizabilities": (C.c_int, [_P, _P, C.c_int64, _P]),
    "rn_potgnn_calc_polarizabilities_f64": (C.c_int, [_P, _P, C.c_int64, _P]),
    "rn_potgnn_calc_polarizabilities_to_device": (C.c_int, [_P, _P, C.c_int64, _P, _P]),
    "rn_potgnn_forward_device": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, C.c_int]),
    "rn_potgnn_forward_device_f64": (C.c_int, [_P, _P, C.c_int64, _P, _P, C.c_int]),
    "rn_potgnn_calc_polarizabilities_async": (C.c_int, [_P, _P, C.c_int64, _P]),
    "rn_potgnn_wait": (C.c_int, [_P]),
    "rn_host_buffer_alloc": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(_P)]),
    "rn_host_buffer_free": (None, [_P]),
    "rn_potgnn_forward": (C.c_int, [_P, _P, C.c_int64, _P]),
    "rn_potgnn_forward_lattices": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "rn_potgnn_forward_samples": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P]),
    "rn_potgnn_forward_samples_f64": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P]),
    "rn_potgnn_raman_tensors": (C.c_int, [_P, _P, _P, C.c_int64, C.c_double, _P]),
    "rn_potgnn_alpha_jacobian": (C.c_int, [_P, _P, C.c_int, _P]),
    "rn_potgnn_raman_tensors_analytic": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "rn_potgnn_set_weights": (C.c_int, [_P, _P, C.c_size_t]),
    "rn_potgnn_train_forward": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P]),
    "rn_potgnn_train_backward": (C.c_int, [_P, _P, _P]),
    "rn_potgnn_set_device_training": (C.c_int, [_P, C.c_int]),
    "rn_potgnn_train_backward_device": (C.c_int, [_P, _P]),
    "rn_potgnn_gradient_buffer": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "rn_potgnn_adam_step": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64]),
    "rn_potgnn_get_weights": (C.c_int, [_P, _P, C.c_size_t]),
    "rn_potgnn_train_forward_f64": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P]),
    "rn_potgnn_train_forward_samples": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "rn_potgnn_train_forward_samples_f64": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, _P, _P]),
    "rn_potgnn_forward_samples_device": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, _P, C.c_int]),
    "rn_potgnn_train_forward_samples_device": (C.c_int, [_P, _P, _P, _P, C.c_int64, _P, _P]),
    "rn_potgnn_train_backward_samples_device": (C.c_int, [_P, _P, _P]),
    "rn_potgnn_train_backward_f64": (C.c_int, [_P, _P, _P]),
    "rn_potgnn_num_triplets": (C.c_int64, [_P]),
    "rn_potgnn_debug_triplets": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "rn_potgnn_debug_ps_schedule": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "rn_potgnn_debug_stage": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t,
                                        C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rn_potgnn_set_profiling": (C.c_int, [_P, C.c_int]),
    "rn_potgnn_kernel_times": (C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                         C.POINTER(C.c_int64), C.c_int]),
    "rn_potgnn_last_error": (C.c_char_p, [_P]),
    "rn_potgnn_version": (C.c_char_p, []),
    "rn_md_raman_intensities": (C.c_int, [_P, C.c_int64, C.c_int, _P, C.c_int64]),
    "rn_md_raman_intensities_device": (C.c_int, [_P, C.c_int64, C.c_int, _P, C.c_int64, _P]),
    # include/rn_ingest.h (host-only trajectory reader)
    "rn_xdatcar_open": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "rn_xdatcar_close": (None, [_P]),
    "rn_xdatcar_info": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int32), _P, C.POINTER(C.c_int32)]),
    "rn_xdatcar_species": (C.c_int, [_P, C.c_int32, C.c_char_p, C.POINTER(C.c_int32)]),
    "rn_xdatcar_read": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, C.c_int]),
    "rn_xdatcar_last_error": (C.c_char_p, [_P]),
    "rn_vasprun_open": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "rn_vasprun_close": (None, [_P]),
    "rn_vasprun_info": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "rn_vasprun_read": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int]),
    "rn_vasprun_timestep": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "rn_vasprun_initial_num_atoms": (C.c_int64, [_P]),
    "rn_vasprun_initial_structure": (C.c_int, [_P, C.POINTER(C.c_int32), _P, _P, C.c_int64, _P, C.c_int64]),
    "rn_vasprun_last_error": (C.c_char_p, [_P]),
}

_lib = None


def library_path() -> str:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load the shared library (once) and attach signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise DeviceError(
            f"{_LIB_PATH} not found: build it with ramannoodle_amd/csrc/build.sh "
            "(there is no CPU fallback for the PotGNN evaluation path)"
        )
    try:
        lib = C.CDLL(_LIB_PATH)
    except OSError as exc:
        raise DeviceError(f"cannot load {_LIB_PATH}: {exc}") from exc
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc: int, handle=None, what: str = "device call") -> None:
    """Map a status code to the exception the reference API contract expects."""
    if rc == RN_OK:
        return
    lib = load()
    msg = lib.rn_potgnn_last_error(handle)
    text = msg.decode() if msg else ""
    if rc == RN_ERR_INVALID_ARGUMENT:
        raise ValueError(f"{what}: {text}")
    if rc == RN_ERR_UNSUPPORTED:
        raise NotImplementedError(f"{what}: {text}")
    if rc == RN_ERR_OUT_OF_MEMORY:
        raise MemoryError(f"{what}: {text}")
    raise DeviceError(f"{what} failed ({rc}): {text}")
