"""VASP XDATCAR trajectories (``ramannoodle/io/vasp/xdatcar.py:21-109``) on the native reader.

``read_positions_ts`` / ``read_trajectory`` keep the reference's signatures, results and
``InvalidFileException`` messages; the file is memory-mapped, indexed once and parsed by
``rn_xdatcar_read`` (``include/rn_ingest.h``) frame-parallel instead of line by line in
Python.  ``XdatcarReader`` exposes the chunked form: ``read(first, count, out=...)`` fills a
caller-owned (e.g. pinned) buffer and releases the GIL, so parsing chunk k+1 overlaps the
device evaluating chunk k (``stream_polarizabilities``).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np
from numpy.typing import NDArray

from ramannoodle_amd import _lib
from ramannoodle_amd.exceptions import InvalidFileException

_FILE_NOT_FOUND, _INVALID_FILE = -6, -7


def _pathify(filepath) -> Path:
    try:
        return Path(filepath)
    except TypeError as exc:
        raise TypeError(f"{filepath} cannot be resolved as a filepath") from exc


class XdatcarReader:
    """An opened XDATCAR: header fields and random access to blocks of frames."""

    def __init__(self, filepath) -> None:
        path = _pathify(filepath)
        self._lib = _lib.load()
        handle = C.c_void_p()
        rc = self._lib.rn_xdatcar_open(str(path).encode(), C.byref(handle))
        self._handle = handle
        if rc == _FILE_NOT_FOUND:
            raise FileNotFoundError(2, "No such file or directory", str(path))
        if rc == _INVALID_FILE:
            message = self._lib.rn_xdatcar_last_error(handle).decode()
            self.close()
            raise InvalidFileException(message)
        if rc != 0:
            raise ValueError(f"rn_xdatcar_open failed with status {rc}")
        frames, atoms, species = C.c_int64(), C.c_int32(), C.c_int32()
        lattice = np.empty((3, 3), dtype=np.float64)
        self._lib.rn_xdatcar_info(handle, C.byref(frames), C.byref(atoms), C.c_void_p(lattice.ctypes.data),
                                  C.byref(species))
        self.num_frames, self.num_atoms, self.lattice = frames.value, atoms.value, lattice
        self.atomic_symbols: list[str] = []
        for k in range(species.value):
            symbol, count = C.create_string_buffer(8), C.c_int32()
            self._lib.rn_xdatcar_species(handle, k, symbol, C.byref(count))
            self.atomic_symbols += [symbol.value.decode()] * count.value

    def read(self, first: int = 0, count: int | None = None, out: NDArray[np.float64] | None = None,
             num_threads: int = 0) -> NDArray[np.float64]:
        """Frames ``[first, first+count)`` as ``float64 (count, N, 3)`` fractional positions,
        exactly as the reference returns them (Cartesian frames converted with
        ``positions @ inv(lattice)``, ``poscar.py:119-120``; no wrapping)."""
        count = self.num_frames - first if count is None else count
        shape = (count, self.num_atoms, 3)
        if out is None:
            out = np.empty(shape, dtype=np.float64)
        elif out.shape != shape or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous float64 array of shape {shape}")
        cartesian = np.zeros(max(count, 1), dtype=np.uint8)
        rc = self._lib.rn_xdatcar_read(self._handle, first, count, C.c_void_p(out.ctypes.data),
                                       C.c_void_p(cartesian.ctypes.data), num_threads)
        if rc == _INVALID_FILE:
            raise InvalidFileException(self._lib.rn_xdatcar_last_error(self._handle).decode())
        if rc != 0:
            raise ValueError(f"rn_xdatcar_read failed with status {rc} (frames {first}..{first + count})")
        if cartesian[:count].any():
            inverse = np.linalg.inv(self.lattice)
            for k in np.nonzero(cartesian[:count])[0]:
                out[k] = out[k] @ inverse
        return out

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.rn_xdatcar_close(self._handle)
            self._handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()


def read_positions_ts(filepath) -> NDArray[np.float64]:
    """Fractional positions time series ``(S, N, 3)`` of an XDATCAR file.

    Raises ``FileNotFoundError`` / ``InvalidFileException`` like the reference."""
    with XdatcarReader(filepath) as reader:
        if reader.num_frames == 0:  # np.array([]) in the reference
            return np.array([])
        return reader.read()


def read_trajectory(filepath, timestep: float):
    """``Trajectory`` from an XDATCAR file; the timestep (fs) is not in the file."""
    from ramannoodle_amd.dynamics import Trajectory
    return Trajectory(read_positions_ts(filepath), timestep)


def stream_polarizabilities(model, filepath, chunk_frames: int = 2000) -> NDArray[np.float64]:
    """Polarizabilities ``(S, 3, 3)`` of every frame of the file without holding the trajectory
    in memory: a worker thread parses block k+1 (the native reader releases the GIL) while the
    device evaluates block k; with the device model the blocks go through page-locked buffers and
    the pipelined entry point (``ramannoodle_amd.io._stream``).  Positions are wrapped into the
    cell as ``Trajectory`` does."""
    from ramannoodle_amd.io._stream import stream_polarizabilities as _stream
    with XdatcarReader(filepath) as reader:
        return _stream(model, reader, chunk_frames)
