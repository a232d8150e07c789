"""The three abstract contracts that form the drop-in boundary
(``ramannoodle/abstract.py:10-83``)."""
from __future__ import annotations

from abc import ABC, abstractmethod

import numpy as np
from numpy.typing import NDArray


class PolarizabilityModel(ABC):
    """Maps fractional positions ``(S,N,3)`` to polarizabilities ``(S,3,3)``."""

    @abstractmethod
    def calc_polarizabilities(self, positions_batch: NDArray[np.float64]) -> NDArray[np.float64]:
        """Return polarizabilities with shape ``(S,3,3)``."""


class RamanSpectrum(ABC):
    """A raw Raman spectrum."""

    @abstractmethod
    def measure(
        self,
        orientation: str | NDArray[np.float64] = "polycrystalline",
        laser_correction: bool = False,
        laser_wavelength: float = 522,
        bose_einstein_correction: bool = False,
        temperature: float = 300,
    ) -> tuple[NDArray[np.float64], NDArray[np.float64]]:
        """Return ``(wavenumbers, intensities)``."""


class Dynamics(ABC):
    """Atomic dynamics from which a Raman spectrum can be computed."""

    @abstractmethod
    def get_raman_spectrum(self, polarizability_model: PolarizabilityModel) -> RamanSpectrum:
        """Compute a Raman spectrum with ``polarizability_model``."""
