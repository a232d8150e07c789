#!/usr/bin/env python3
"""Exercise the remaining BASELINE.json configurations on one MI355X and print timings:
config 3 share (256 atoms, 1250 frames = one GPU's block of the 10k-frame trajectory),
config 4 (256 atoms, 768 phonon modes, +-delta finite differences in float64),
and the documented parity hyper-parameters (Fn=5, Fe=14)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_workload  # noqa: E402
from ramannoodle_amd.dynamics import Phonons, Trajectory  # noqa: E402

out = {}


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t) / reps


# ---- config 3 share
wl = make_workload((4, 4, 2), 1250, "perf", seed=33)
model = wl["model"]()
pos = torch.tensor(wl["positions"], device="cuda")
res, dt = timed(lambda: model.calc_polarizabilities_device(pos, synchronize=True))
out["config3_share"] = {"atoms": model.num_atoms, "edges": model.num_edges, "frames": 1250,
                        "structures_per_s": 1250 / dt}
spec, dt_host = timed(lambda: Trajectory(wl["positions"], 1.0).get_raman_spectrum(model), reps=1)
w, i = spec.measure(bose_einstein_correction=True, temperature=300)
out["config3_share"]["host_api_structures_per_s"] = 1250 / dt_host
out["config3_share"]["md_spectrum_bins"] = int(w.size)

# ---- config 4: phonon Raman, 768 modes, float64 finite differences
rng = np.random.default_rng(44)
n = model.num_atoms
qmat, _ = np.linalg.qr(rng.normal(size=(3 * n, 3 * n)))
zs = np.array([12 if k % 2 == 0 else 8 for k in range(n)])  # masses only scale the synthetic modes
mass = np.where(zs == 12, 24.305, 15.999)
disp = (qmat.T.reshape(3 * n, n, 3) / np.sqrt(mass)[None, :, None]) / np.diag(wl["lattice"])[None, None, :]
wn = np.linspace(50.0, 900.0, 3 * n)
ref = wl["positions"][0]  # a thermally displaced frame: the ideal sites are inversion centres
                           # (first-order Raman forbidden, all tensors vanish)
ph = Phonons(ref, wn, disp)
t = time.perf_counter()
spec = ph.get_raman_spectrum(model)
dt = time.perf_counter() - t
wv, inten = spec.measure(laser_correction=True, laser_wavelength=532)
out["config4"] = {"modes": 3 * n, "displaced_cells": 6 * n, "seconds": dt,
                  "structures_per_s_f64": 6 * n / dt,
                  "raman_tensor_rms": float(np.sqrt((spec.raman_tensors ** 2).mean()))}
t = time.perf_counter()
analytic = model.calc_raman_tensors(ref, disp, method="analytic")
out["config4"]["analytic_seconds"] = time.perf_counter() - t
out["config4"]["analytic_vs_fd_rel"] = float(
    np.abs(analytic - spec.raman_tensors).max() / np.abs(spec.raman_tensors).max())
# fp32 finite differences for comparison (the reference's default precision): noise level
plus = model.calc_polarizabilities(ref[None] + disp[:8] * 1e-3)
minus = model.calc_polarizabilities(ref[None] - disp[:8] * 1e-3)
fd32 = (plus - minus) / 1e-3
out["config4"]["fp32_fd_rel_error_vs_f64"] = float(
    np.abs(fd32 - spec.raman_tensors[:8]).max() / np.abs(spec.raman_tensors[:8]).max())

# ---- parity hyper-parameters (the only documented set), 128 atoms
wl = make_workload((4, 2, 2), 1000, "parity", seed=22)
model = wl["model"]()
pos = torch.tensor(wl["positions"], device="cuda")
res, dt = timed(lambda: model.calc_polarizabilities_device(pos, synchronize=True))
out["parity_hparams_128atoms"] = {"structures_per_s": 1000 / dt}
print(json.dumps(out, indent=1))

# ---- config 5 (training): forward + backward + Adam step per batch, synthetic teacher targets
from ramannoodle_amd.dataset import PolarizabilityDataset  # noqa: E402
from ramannoodle_amd.pmodel import train_single_epoch  # noqa: E402

from bench import rocksalt  # noqa: E402
from ramannoodle_amd.pmodel import DeviceAdam  # noqa: E402

# BASELINE config 5 is the 256-atom cell; (the 128-atom lines stay for continuity with round 1)
CASES = (("perf", (4, 4, 2), 256, 32, "device"), ("perf", (4, 4, 2), 256, 32, "host"),
         ("parity", (4, 4, 2), 512, 32, "device"), ("perf", (4, 2, 2), 128, 32, "host"))
if os.environ.get("RN_CONFIG5_FULL") == "1":  # BASELINE config 5 at its full size: 50 000 structures per epoch
    CASES = (("perf", (4, 4, 2), 50000, 32, "device"),) + CASES
for hp, cells, frames, batch, where in CASES:
    wl = make_workload(cells, frames, hp, seed=55)
    teacher = wl["model"]()
    alpha = teacher.calc_polarizabilities(wl["positions"])
    lattice, ref_pos, zs = rocksalt(*cells)
    ds = PolarizabilityDataset(lattice, zs, wl["positions"], alpha)
    student = wl["model"]()
    # "device": weights, gradients and Adam moments stay in HBM; "host": torch.optim.Adam on the host copy
    opt = DeviceAdam(student, lr=1e-3) if where == "device" else torch.optim.Adam(student.parameters(), lr=1e-3)
    small = frames <= 1024
    val = ds if small else torch.utils.data.Subset(ds, range(2000))
    warm = ds if small else torch.utils.data.Subset(ds, range(256))
    train_single_epoch(student, warm, warm, batch, opt, torch.nn.MSELoss())  # warm-up
    t = time.perf_counter()
    losses = train_single_epoch(student, ds, val, batch, opt, torch.nn.MSELoss())
    dt = time.perf_counter() - t
    out[f"config5_training_{hp}_{teacher.num_atoms}atoms_{frames}structures_{where}_adam"] = {
        "atoms": teacher.num_atoms, "structures": frames, "batch": batch, "epoch_seconds": dt,
        "note": ("one epoch = training pass + validation pass over the same structures" if small
                 else "one epoch = training pass over all structures + validation pass over 2000 of them"),
        "train_structures_per_s": frames / dt, "train_loss": losses[0]}
print(json.dumps({k: v for k, v in out.items() if k.startswith("config5")}, indent=1))
