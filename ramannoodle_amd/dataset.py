"""Polarizability dataset for training (``ramannoodle/dataset/torch/_dataset.py``)."""
from __future__ import annotations

import numpy as np
import torch
from numpy.typing import NDArray
from torch.utils.data import Dataset

from ramannoodle_amd.exceptions import get_type_error, verify_ndarray_shape
from ramannoodle_amd.pmodel.potgnn import polarizability_tensors_to_vectors


def scale_and_flatten_polarizabilities(polarizabilities: torch.Tensor, scale_mode: str):
    """Element-wise mean, population standard deviation and the scaled 6-vectors
    ``(xx,yy,zz,xy,xz,yz)`` (``_dataset.py:21-62``).  ``scale_mode``: ``"standard"``
    ``(a-mean)/std``, ``"stddev"`` ``(a-mean)/std+mean`` or ``"none"``."""
    mean = polarizabilities.mean(0, keepdim=True)
    stddev = polarizabilities.std(0, unbiased=False, keepdim=True)
    if scale_mode == "standard":
        polarizabilities = (polarizabilities - mean) / stddev
    elif scale_mode == "stddev":
        polarizabilities = (polarizabilities - mean) / stddev + mean
    elif scale_mode != "none":
        raise ValueError(f"unsupported scale mode: {scale_mode}")
    return mean, stddev, polarizability_tensors_to_vectors(polarizabilities)


class PolarizabilityDataset(Dataset):
    """Structures (fractional positions ``(S,N,3)`` of one cell) with polarizabilities
    ``(S,3,3)``, standard-scaled and flattened to 6-vectors (``_dataset.py:65-230``)."""

    def __init__(self, lattice: NDArray[np.float64], atomic_numbers: list[int],
                 positions: NDArray[np.float64], polarizabilities: NDArray[np.float64]):
        verify_ndarray_shape("lattice", lattice, (3, 3))
        if not isinstance(atomic_numbers, list):
            raise get_type_error("atomic_numbers", atomic_numbers, "list")
        num_atoms = len(atomic_numbers)
        verify_ndarray_shape("positions", positions, (None, num_atoms, 3))
        num_samples = positions.shape[0]
        verify_ndarray_shape("polarizabilities", polarizabilities, (num_samples, 3, 3))
        dtype = torch.get_default_dtype()
        self._lattices = torch.tensor(lattice).type(dtype).unsqueeze(0).expand(num_samples, 3, 3)
        self._atomic_numbers = torch.tensor(atomic_numbers).type(torch.int).unsqueeze(0).expand(
            num_samples, num_atoms)
        self._positions = torch.tensor(positions).type(dtype)
        self._polarizabilities = torch.tensor(polarizabilities)
        _, _, scaled = scale_and_flatten_polarizabilities(self._polarizabilities, "standard")
        self._scaled_polarizabilities = scaled.type(dtype)

    @property
    def num_atoms(self) -> int:
        return self._positions.size(1)

    @property
    def num_samples(self) -> int:
        return self._positions.size(0)

    @property
    def atomic_numbers(self) -> list[int]:
        return [int(n) for n in self._atomic_numbers[0]]

    @property
    def positions(self) -> NDArray[np.float64]:
        return self._positions.detach().clone().numpy()

    @property
    def polarizabilities(self) -> NDArray[np.float64]:
        return self._polarizabilities.detach().clone().numpy()

    @property
    def scaled_polarizabilities(self) -> NDArray[np.float64]:
        return self._scaled_polarizabilities.detach().clone().numpy()

    @property
    def mean_polarizability(self) -> NDArray[np.float64]:
        return self._polarizabilities.mean(0).clone().numpy()

    @property
    def stddev_polarizability(self) -> NDArray[np.float64]:
        return self._polarizabilities.std(0, unbiased=False).clone().numpy()

    def scale_polarizabilities(self, mean: NDArray[np.float64], stddev: NDArray[np.float64]) -> None:
        """Re-standardise with the statistics of another (training) set (``:193-217``)."""
        verify_ndarray_shape("mean", mean, (3, 3))
        verify_ndarray_shape("mean", stddev, (3, 3))
        scaled = (self._polarizabilities.detach().clone() - torch.tensor(mean)) / torch.tensor(stddev)
        self._scaled_polarizabilities = polarizability_tensors_to_vectors(scaled)

    def __len__(self) -> int:
        return self.num_samples

    def __getitem__(self, i: int):
        return (self._lattices[i], self._atomic_numbers[i], self._positions[i],
                self._scaled_polarizabilities[i])
