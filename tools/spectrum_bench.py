#!/usr/bin/env python3
"""Host (numpy/scipy) vs device (rn_md_raman_intensities) reduction of a polarizability time series."""
import os, sys, time
import numpy as np
import torch  # before the HIP library: one HIP runtime per process (torch's)
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ramannoodle_amd.spectrum import MDRamanSpectrum
rng = np.random.default_rng(0)
MDRamanSpectrum(rng.normal(size=(64, 3, 3)), 1.0).measure(device=0)  # loads hipFFT
for steps in (10_000, 100_000, 1_000_000):
    a = rng.normal(size=(steps, 3, 3)); a = a + a.transpose(0, 2, 1)
    sp = MDRamanSpectrum(a, 1.0)
    t = time.perf_counter(); w, ih = sp.measure(); th = time.perf_counter() - t
    sp.measure(device=0)
    t = time.perf_counter(); w, idv = sp.measure(device=0); td = time.perf_counter() - t
    from ramannoodle_amd.spectrum import DeviceMDRamanSpectrum
    dev = DeviceMDRamanSpectrum(torch.tensor(a, device="cuda"), 1.0)
    dev.measure()
    torch.cuda.synchronize()
    t = time.perf_counter(); w, idr = dev.measure(); tr = time.perf_counter() - t
    print(f"S = {steps:8d}: host {th * 1e3:8.1f} ms   device from host alpha {td * 1e3:7.2f} ms (incl. H2D/D2H; plans cached)   "
          f"device-resident alpha {tr * 1e3:7.2f} ms   max rel diff {np.abs(idv - ih).max() / np.abs(ih).max():.1e} / "
          f"{np.abs(idr - ih).max() / np.abs(ih).max():.1e}")
