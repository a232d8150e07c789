"""Minimal structure support for the evaluation path.

PotGNN reads only four attributes of the reference's ``ReferenceStructure``
(``lattice``, ``positions``, ``atomic_numbers``, ``num_atoms``:
``ramannoodle/pmodel/torch/_gnn.py:490-498,502,684,698-701``); symmetry handling is out
of scope.  Any object with those attributes works.
"""
from __future__ import annotations

import numpy as np
from numpy.typing import NDArray

from ramannoodle_amd.exceptions import get_type_error, verify_ndarray_shape


def apply_pbc(positions: NDArray[np.float64]) -> NDArray[np.float64]:
    """Wrap fractional coordinates into [0, 1) (``ramannoodle/structure/utils.py:13-29``: ``x - x // 1``).

    For floating-point arrays ``x // 1`` is ``floor(x)`` bit for bit (also for -0.0, infinities and NaN), and numpy's
    ``floor_divide`` is fourteen times slower than ``floor`` (41 ms against 3 ms for a 1250-frame, 256-atom block: twice
    the time the GPU needs to evaluate those frames), so arrays take ``floor``; everything else takes the reference's
    expression, with its exceptions."""
    if isinstance(positions, np.ndarray) and positions.dtype.kind == "f":
        wrapped = np.floor(positions)
        return np.subtract(positions, wrapped, out=wrapped)  # (one temporary, not two: the arrays are trajectories)
    try:
        return positions - positions // 1
    except TypeError as exc:
        raise get_type_error("positions", positions, "ndarray") from exc


class ReferenceStructure:
    """Lattice (rows = vectors, angstrom), fractional positions and atomic numbers."""

    def __init__(self, atomic_numbers: list[int], lattice: NDArray[np.float64],
                 positions: NDArray[np.float64]) -> None:
        if not isinstance(atomic_numbers, list):
            raise get_type_error("atomic_numbers", atomic_numbers, "list")
        verify_ndarray_shape("lattice", lattice, (3, 3))
        verify_ndarray_shape("positions", positions, (len(atomic_numbers), 3))
        self._atomic_numbers = [int(z) for z in atomic_numbers]
        self._lattice = np.array(lattice, dtype=np.float64)
        self._positions = np.array(positions, dtype=np.float64)

    @property
    def atomic_numbers(self) -> list[int]:
        return list(self._atomic_numbers)

    @property
    def num_atoms(self) -> int:
        return len(self._atomic_numbers)

    @property
    def lattice(self) -> NDArray[np.float64]:
        return self._lattice.copy()

    @property
    def positions(self) -> NDArray[np.float64]:
        return self._positions.copy()
