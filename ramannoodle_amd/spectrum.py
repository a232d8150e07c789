"""Raman spectra from Raman tensors (phonons) or a polarizability time series (MD).

Host-side numpy/scipy post-processing of the device results, kept on the CPU as the
scope table prescribes (ms-scale even at 10^4 frames).
Follows ``ramannoodle/spectrum/_raman.py`` and ``ramannoodle/spectrum/utils.py``.
"""
from __future__ import annotations

import numpy as np
import scipy.fftpack
import scipy.signal
from numpy.typing import NDArray

from ramannoodle_amd.abstract import RamanSpectrum
from ramannoodle_amd.constants import BOLTZMANN_CONSTANT
from ramannoodle_amd.exceptions import get_type_error, verify_ndarray_shape

_CM1_TO_HZ = 29979245800.0
_PLANCK_EV_S = 4.1357e-15  # value used by the reference (spectrum/_raman.py:37)
_PER_FS_TO_CM1 = 33.35640951981521 * 1e3  # spectrum/utils.py:118-121


def get_bose_einstein_correction(wavenumbers: NDArray[np.float64],
                                 temperature: float) -> NDArray[np.float64]:
    """``1 / (1 - exp(-E/kT))`` (``spectrum/_raman.py:13-40``)."""
    try:
        if temperature <= 0:
            raise ValueError(f"invalid temperature: {temperature} <= 0")
    except TypeError as exc:
        raise get_type_error("temperature", temperature, "float") from exc
    try:
        energy = wavenumbers * _CM1_TO_HZ * _PLANCK_EV_S
        return 1 / (1 - np.exp(-energy / (BOLTZMANN_CONSTANT * temperature)))
    except TypeError as exc:
        raise get_type_error("wavenumbers", wavenumbers, "ndarray") from exc


def get_laser_correction(wavenumbers: NDArray[np.float64],
                         laser_wavenumber: float) -> NDArray[np.float64]:
    """``((nu - nu_L)/1e4)^4 / nu`` (``spectrum/_raman.py:43-69``)."""
    try:
        if laser_wavenumber <= 0:
            raise ValueError(f"invalid laser_wavenumber: {laser_wavenumber} <= 0")
    except TypeError as exc:
        raise get_type_error("laser_wavenumber", laser_wavenumber, "float") from exc
    try:
        return ((wavenumbers - laser_wavenumber) / 10000) ** 4 / wavenumbers
    except TypeError as exc:
        raise get_type_error("wavenumbers", wavenumbers, "ndarray") from exc


def _apply_corrections(wavenumbers, intensities, laser_correction, laser_wavelength,
                       bose_einstein_correction, temperature):
    if laser_correction:
        intensities = intensities * get_laser_correction(wavenumbers, 10000000 / laser_wavelength)
    if bose_einstein_correction:
        intensities = intensities * get_bose_einstein_correction(wavenumbers, temperature)
    return intensities


def _require_polycrystalline(orientation) -> None:
    if not (isinstance(orientation, str) and orientation == "polycrystalline"):
        raise NotImplementedError("only polycrystalline spectra are supported for now")


def calc_signal_spectrum(signal: NDArray[np.float64],
                         sampling_rate: float) -> tuple[NDArray[np.float64], NDArray[np.float64]]:
    """Non-negative-frequency FFT of the positive-lag autocorrelation
    (``spectrum/utils.py:76-124``)."""
    verify_ndarray_shape("signal", signal, (None,))
    full = scipy.signal.correlate(signal, signal, "full")
    autocorrelation = full[(len(full) - 1) // 2:]
    wavenumbers = scipy.fftpack.fftfreq(autocorrelation.size, sampling_rate) * _PER_FS_TO_CM1
    intensities = np.real(scipy.fftpack.fft(autocorrelation))
    keep = wavenumbers >= 0
    return wavenumbers[keep], intensities[keep]


def convolve_spectrum(wavenumbers, intensities, function: str = "gaussian", width: float = 5,
                      out_wavenumbers=None):
    """Gaussian / Lorentzian broadening (``spectrum/utils.py:13-73``)."""
    if out_wavenumbers is None:
        lo, hi = np.min(wavenumbers) - 100, np.max(wavenumbers) + 100
        out_wavenumbers = np.linspace(lo, hi, int(np.rint(hi - lo)))
    verify_ndarray_shape("out_wavenumbers", out_wavenumbers, (None,))
    verify_ndarray_shape("wavenumbers", wavenumbers, (None,))
    verify_ndarray_shape("intensities", intensities, (len(wavenumbers),))
    try:
        if width <= 0:
            raise ValueError(f"invalid width: {width} <= 0")
    except TypeError as exc:
        raise get_type_error("width", width, "float") from exc
    if function not in ("gaussian", "lorentzian"):
        raise ValueError(f"unsupported convolution type: {function}")
    delta = np.asarray(wavenumbers)[:, None] - out_wavenumbers[None, :]
    if function == "gaussian":
        kernel = (1 / width) * (1 / np.sqrt(2 * np.pi)) * np.exp(-(delta**2) / (2 * width**2))
    else:
        kernel = (1 / np.pi) * (0.5 * width / (delta**2 + (0.5 * width) ** 2))
    return out_wavenumbers, (kernel * np.asarray(intensities)[:, None]).sum(axis=0)


def _md_intensities_on_device(polarizability_ts: NDArray[np.float64], timestep: float, device: int):
    """(wavenumbers, uncorrected intensities) of ``MDRamanSpectrum.measure`` from the device path."""
    import ctypes as C

    from ramannoodle_amd import _lib
    alpha = np.ascontiguousarray(polarizability_ts, dtype=np.float64)
    n = alpha.shape[0] - 1
    if n < 2:
        raise ValueError("the device reduction needs at least three time steps")
    bins = (n + 1) // 2 - 1
    intensities = np.empty(bins, dtype=np.float64)
    rc = _lib.load().rn_md_raman_intensities(C.c_void_p(alpha.ctypes.data), alpha.shape[0], device,
                                             C.c_void_p(intensities.ctypes.data), bins)
    _lib.check(rc, None, "rn_md_raman_intensities")
    wavenumbers = scipy.fftpack.fftfreq(n, timestep) * _PER_FS_TO_CM1
    return wavenumbers[1:bins + 1], intensities


def _md_intensities_device_resident(alpha_device, timestep: float):
    """The same reduction for a time series that already lives in HBM (a torch CUDA tensor
    ``float64[S,3,3]``): ``rn_md_raman_intensities_device`` -- only the intensities reach the host."""
    import ctypes as C

    import torch

    from ramannoodle_amd import _lib
    n = alpha_device.shape[0] - 1
    if n < 2:
        raise ValueError("the device reduction needs at least three time steps")
    bins = (n + 1) // 2 - 1
    intensities = np.empty(bins, dtype=np.float64)
    device = alpha_device.device.index if alpha_device.device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(alpha_device.device).cuda_stream
    rc = _lib.load().rn_md_raman_intensities_device(C.c_void_p(alpha_device.data_ptr()), alpha_device.shape[0],
                                                    device, C.c_void_p(intensities.ctypes.data), bins,
                                                    C.c_void_p(stream))
    _lib.check(rc, None, "rn_md_raman_intensities_device")
    wavenumbers = scipy.fftpack.fftfreq(n, timestep) * _PER_FS_TO_CM1
    return wavenumbers[1:bins + 1], intensities


class PhononRamanSpectrum(RamanSpectrum):
    """First-order spectrum from phonon wavenumbers ``(M,)`` and Raman tensors ``(M,3,3)``
    (``spectrum/_raman.py:72-194``)."""

    def __init__(self, phonon_wavenumbers, raman_tensors) -> None:
        verify_ndarray_shape("phonon_wavenumbers", phonon_wavenumbers, (None,))
        verify_ndarray_shape("raman_tensors", raman_tensors, (len(phonon_wavenumbers), 3, 3))
        self._phonon_wavenumbers = phonon_wavenumbers
        self._raman_tensors = raman_tensors

    @property
    def phonon_wavenumbers(self):
        return self._phonon_wavenumbers.copy()

    @property
    def raman_tensors(self):
        return self._raman_tensors.copy()

    def measure(self, orientation="polycrystalline", laser_correction=False,
                laser_wavelength=522, bose_einstein_correction=False, temperature=300):
        _require_polycrystalline(orientation)
        r = self._raman_tensors
        xx, yy, zz = r[:, 0, 0], r[:, 1, 1], r[:, 2, 2]
        alpha_squared = ((xx + yy + zz) / 3.0) ** 2
        gamma_squared = (
            (xx - yy) ** 2 + (xx - zz) ** 2 + (yy - zz) ** 2
            + 6.0 * (r[:, 0, 1] ** 2 + r[:, 0, 2] ** 2 + r[:, 1, 2] ** 2)
        ) / 2.0
        intensities = 45.0 * alpha_squared + 7.0 * gamma_squared
        intensities = _apply_corrections(self._phonon_wavenumbers, intensities, laser_correction,
                                         laser_wavelength, bose_einstein_correction, temperature)
        return self._phonon_wavenumbers, intensities


class MDRamanSpectrum(RamanSpectrum):
    """Spectrum from a polarizability time series ``(S,3,3)`` and a timestep in fs
    (``spectrum/_raman.py:197-309``)."""

    def __init__(self, polarizability_ts, timestep: float):
        verify_ndarray_shape("polarizability_ts", polarizability_ts, (None, 3, 3))
        self._polarizability_ts = polarizability_ts
        self._timestep = timestep

    @property
    def polarizability_ts(self):
        return self._polarizability_ts

    @property
    def timestep(self) -> float:
        return self._timestep

    def _measure_on_device(self, device: int):
        return _md_intensities_on_device(self._polarizability_ts, self._timestep, device)

    def measure(self, orientation="polycrystalline", laser_correction=False,
                laser_wavelength=522, bose_einstein_correction=False, temperature=300, device=None):
        """``device`` (an int, not in the reference's signature): reduce the time series on that
        GPU (``rn_md_raman_intensities``: one batched FFT, one power spectrum, two more FFTs)
        instead of on the host; pays off for very long series (S >> 1e5) or many models."""
        _require_polycrystalline(orientation)
        dt = self._timestep
        if device is not None:
            wavenumbers, intensities = self._measure_on_device(int(device))
            intensities = _apply_corrections(wavenumbers, intensities, laser_correction,
                                             laser_wavelength, bose_einstein_correction, temperature)
            return wavenumbers, intensities
        ad = np.diff(self._polarizability_ts, axis=0)  # d(alpha)/dt up to a constant

        def spec(sig):
            return calc_signal_spectrum(sig, dt)[1]

        wavenumbers, _ = calc_signal_spectrum(ad[:, 0, 0], dt)
        alpha2 = (1 / 9) * spec(ad[:, 0, 0] + ad[:, 1, 1] + ad[:, 2, 2])
        gamma2 = (
            (1 / 2) * spec(ad[:, 0, 0] - ad[:, 1, 1])
            + (1 / 2) * spec(ad[:, 1, 1] - ad[:, 2, 2])
            + (1 / 2) * spec(ad[:, 2, 2] - ad[:, 0, 0])
            + 3 * spec(ad[:, 0, 1]) + 3 * spec(ad[:, 1, 2]) + 3 * spec(ad[:, 0, 2])
        )
        intensities = (45.0 * alpha2 + 7.0 * gamma2)[1:]  # the 0 cm^-1 bin is dropped
        wavenumbers = wavenumbers[1:]
        intensities = _apply_corrections(wavenumbers, intensities, laser_correction,
                                         laser_wavelength, bose_einstein_correction, temperature)
        return wavenumbers, intensities


class DeviceMDRamanSpectrum(MDRamanSpectrum):
    """``MDRamanSpectrum`` whose polarizability time series stays where the evaluator wrote it
    (HBM, a contiguous torch CUDA tensor ``float64[S,3,3]``): ``measure`` reduces it on that GPU
    and only the intensities travel to the host.  ``polarizability_ts``
    copies the series to the host on first use, for callers that want the numbers themselves."""

    def __init__(self, polarizability_ts_device, timestep: float):  # pylint: disable=super-init-not-called
        shape = tuple(polarizability_ts_device.shape)
        if len(shape) != 3 or shape[1:] != (3, 3):
            raise ValueError(f"polarizability_ts has wrong shape: {shape} != (_,3,3)")
        if not (polarizability_ts_device.is_cuda and polarizability_ts_device.is_contiguous()
                and str(polarizability_ts_device.dtype) == "torch.float64"):
            raise ValueError("polarizability_ts must be a contiguous float64 CUDA tensor")
        self._device_ts = polarizability_ts_device
        self._host_ts = None
        self._timestep = timestep

    @property
    def polarizability_ts(self):
        if self._host_ts is None:
            self._host_ts = self._device_ts.cpu().numpy()
        return self._host_ts

    @property
    def _polarizability_ts(self):  # what the host path of MDRamanSpectrum.measure reads
        return self.polarizability_ts

    def _measure_on_device(self, device: int):
        if device != (self._device_ts.device.index or 0):
            return _md_intensities_on_device(self.polarizability_ts, self._timestep, device)
        return _md_intensities_device_resident(self._device_ts, self._timestep)

    def measure(self, orientation="polycrystalline", laser_correction=False, laser_wavelength=522,
                bose_einstein_correction=False, temperature=300, device=None, host=False):
        """As ``MDRamanSpectrum.measure``; reduces on the tensor's GPU unless ``host=True``."""
        if device is None and not host:
            device = self._device_ts.device.index or 0
        return super().measure(orientation, laser_correction, laser_wavelength, bose_einstein_correction,
                               temperature, device=None if host else device)
