"""Streamed evaluation of a trajectory file: parse block k+1 on a worker
thread (the native readers release the GIL) while the device evaluates block k.

With the device model (``PotGNN``) the blocks are parsed straight into page-locked buffers
(``rn_host_buffer_alloc``) and handed to ``rn_potgnn_calc_polarizabilities_async``: the
host-to-device copy of block k+1 runs on the handle's copy stream while block k is still being
evaluated, and results come back into a page-locked array.  Any other ``PolarizabilityModel`` is
called block by block through ``calc_polarizabilities``.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np
import torch

from ramannoodle_amd import _lib


class PinnedArray:
    """A float64 numpy view of page-locked host memory (``hipHostMalloc``)."""

    def __init__(self, shape, device: int = 0) -> None:
        self._lib = _lib.load()
        count = int(np.prod(shape))
        ptr = C.c_void_p()
        rc = self._lib.rn_host_buffer_alloc(max(count, 1) * 8, int(device), C.byref(ptr))
        _lib.check(rc, None, "rn_host_buffer_alloc")
        self._ptr = ptr
        self.array = np.ctypeslib.as_array((C.c_double * max(count, 1)).from_address(ptr.value))[:count].reshape(shape)

    def free(self) -> None:
        if getattr(self, "_ptr", None):
            self.array = None
            self._lib.rn_host_buffer_free(self._ptr)
            self._ptr = None

    def __del__(self):
        self.free()


def _wrap_in_place(block: np.ndarray) -> None:
    """``apply_pbc`` (``x - x // 1``, ``structure/utils.py:13-29``) without a temporary copy."""
    np.subtract(block, np.floor(block), out=block)  # (floor(x) == x // 1 bit for bit, at a fourteenth of the time)


def stream_polarizabilities(model, reader, chunk_frames: int = 2000) -> np.ndarray:
    """Polarizabilities ``(S,3,3)`` of every frame ``reader`` (an ``XdatcarReader`` /
    ``VasprunReader``) holds, wrapped into the cell as ``Trajectory`` does."""
    total, atoms = reader.num_frames, reader.num_atoms
    bounds = [(lo, min(lo + chunk_frames, total)) for lo in range(0, total, chunk_frames)]
    # the pipelined entry evaluates in float32; under torch.set_default_dtype(float64) the evaluation follows the
    # default dtype as calc_polarizabilities does (_gnn.py:705-710), block by block through the synchronous call
    pipelined = hasattr(model, "calc_polarizabilities_async") and torch.get_default_dtype() != torch.float64
    pinned = []
    if pipelined:
        device = model.device_index
        # FOUR input buffers: while block k is submitted, the worker parses block k+1 into the buffer
        # block k-3 used.  calc_polarizabilities_async(k-1) returned only after the staging slot it
        # shares with block k-3 (two slots, alternating) had finished -- copy-in included -- so that
        # buffer is free by construction; with three buffers the worker would write block k-2's
        # buffer before anything had waited for block k-2's host-to-device copy.
        pinned = [PinnedArray((chunk_frames, atoms, 3), device) for _ in range(4)]
        out_pin = PinnedArray((total, 3, 3), device)
        pinned.append(out_pin)
        buffers, result = [p.array for p in pinned[:4]], out_pin.array
    else:
        buffers = [np.empty((chunk_frames, atoms, 3), dtype=np.float64) for _ in range(2)]
        result = np.empty((total, 3, 3), dtype=np.float64)
    errors: list[BaseException] = []

    def parse(k):
        lo, hi = bounds[k]
        try:
            block = buffers[k % len(buffers)][: hi - lo]
            reader.read(lo, hi - lo, out=block)
            _wrap_in_place(block)
        except BaseException as exc:  # pylint: disable=broad-except  (re-raised by the caller)
            errors.append(exc)

    try:
        if bounds:
            parse(0)
        for k, (lo, hi) in enumerate(bounds):
            if errors:
                raise errors[0]
            worker = None
            if k + 1 < len(bounds):
                # (pipelined: blocks k-1 and k-2 may still be in flight on the device while block k is
                #  submitted and block k+1 is parsed -- see the buffer count above)
                worker = threading.Thread(target=parse, args=(k + 1,))
                worker.start()
            block = buffers[k % len(buffers)][: hi - lo]
            if pipelined:
                model.calc_polarizabilities_async(block, result[lo:hi])
            else:
                result[lo:hi] = model.calc_polarizabilities(block)
            if worker is not None:
                worker.join()
        if pipelined:
            model.wait()
        if errors:
            raise errors[0]
        return np.array(result)  # (a copy: the page-locked staging memory is released below)
    finally:
        if pipelined:
            try:
                model.wait()
            finally:
                for p in pinned:
                    p.free()
