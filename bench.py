#!/usr/bin/env python3
"""Benchmark: structures/s of GNN polarizability evaluation on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config 3|2]

A "step" is one pass of the hot path -- ``PotGNN.calc_polarizabilities`` on synthetic MD
frames already resident in HBM -- plus, for N > 1, the single all-gather of the per-frame
polarizabilities (RCCL).  Rank 0 prints ONE JSON line.

``--config 3`` (default, BASELINE.json configs[2], the north star's workload): 256-atom
rocksalt cell (cutoff 3.2 A: E = 4608 directed edges, T = 78336 triplets), ONE 10 000-frame
MD trajectory, perf widths Fn = Fe = 64, P = 4.  The frames are sharded over the N ranks in
contiguous blocks (1250 per GPU at N = 8): strong scaling, total work fixed.
``--config 2`` (configs[1]): 128-atom cell, 1000 frames PER GPU (weak scaling).

Launch: with ``--gpus N > 1`` and no ``WORLD_SIZE`` in the environment this process starts
the N ranks itself (plain child processes, started before anything here touches the GPU)
and exits with their status; under ``torch.distributed.run`` (``WORLD_SIZE`` set) it is one
rank, and ``--gpus`` must equal ``WORLD_SIZE``.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HPARAMS = {
    # name: (Fn, Fe, P)
    "perf": (64, 64, 4),
    "parity": (5, 14, 4),  # the only documented set (machine-learning.ipynb:187-192)
    # rows of exactly one 64-byte line (16 floats): the narrow kernels' algorithmic bytes ARE their HBM lines, so this is
    # the set on which the NodeBlock scatter-aggregate shows what it streams (north star: >= 40 % of the HBM peak)
    "line16": (16, 16, 4),
    "n8e16": (8, 16, 4),
}
EDGE_AGG_KERNEL_ID = 7  # index of "edge_agg" in rn_potgnn_kernel_times / csrc/api.hip
PROJ_C3_KERNEL_ID = 5   # "proj_edge_c3": the [E,64]x[64,256] projection (HBM-bound; unfused pipeline)
PROJ_C1_KERNEL_ID = 3   # "proj_edge_c1": the [E,64]x[64,128] projection (timed with --profile-all only:
                        # event pairs around it cost the two-lane schedule ~3 % of throughput)
NODE_AGG_KERNEL_ID = 4  # "node_agg": the NodeBlock scatter-aggregate (fused: incl. its c1 projection)
LIGHT_CM_PER_FS = 2.99792458e-5


# ----------------------------------------------------------------------------- workload
def rocksalt(nx, ny, nz, a=4.2):
    """Rocksalt supercell: Z=12 on even-parity sites, Z=8 on odd."""
    pos, zs = [], []
    for ix in range(2 * nx):
        for iy in range(2 * ny):
            for iz in range(2 * nz):
                pos.append([ix / (2.0 * nx), iy / (2.0 * ny), iz / (2.0 * nz)])
                zs.append(12 if (ix + iy + iz) % 2 == 0 else 8)
    lattice = np.diag([nx * a, ny * a, nz * a]).astype(np.float64)
    return lattice, np.array(pos), zs


def md_frames(rng, lattice, ref, frames, amplitude=0.05, dt_fs=1.0, t0=0):
    """Bounded per-atom sinusoids (A = 0.05 A, 100-800 cm^-1): no neighbour crosses the
    cutoff shell, and the MD spectrum is structured."""
    n = ref.shape[0]
    nu = rng.uniform(100.0, 800.0, (1, n, 3))
    phi = rng.uniform(0.0, 2 * np.pi, (1, n, 3))
    t = (t0 + np.arange(frames))[:, None, None] * dt_fs
    cart = amplitude * np.cos(2 * np.pi * LIGHT_CM_PER_FS * nu * t + phi)
    x = ref[None] + cart / np.diag(lattice)[None, None, :]
    return x - np.floor(x)


def synthetic_state(model, seed):
    """Notebook-style init (weights ~ N(0,1), Linear bias ~ U(-0.5,0.5)) + non-trivial
    LayerNorm / BatchNorm statistics."""
    gen = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    for k, v in sd.items():
        if not v.is_floating_point() or k == "_edge_embedding.offset":
            continue
        norm = "norm" in k or "_to_polarizability_embedding.1." in k
        if k.endswith("running_mean"):
            v.copy_(torch.randn(v.shape, generator=gen) * 0.3)
        elif k.endswith("running_var"):
            v.copy_(torch.rand(v.shape, generator=gen) + 0.5)
        elif norm and k.endswith("weight"):
            v.copy_(torch.rand(v.shape, generator=gen) + 0.5)
        elif norm and k.endswith("bias"):
            v.copy_(torch.rand(v.shape, generator=gen) * 0.6 - 0.3)
        elif k.endswith("bias"):
            v.copy_(torch.rand(v.shape, generator=gen) - 0.5)
        else:
            v.copy_(torch.randn(v.shape, generator=gen))
    return sd


def make_workload(num_cells=(4, 2, 2), frames=1000, hparams="perf", seed=22, t0=0, structure=None, cutoff=3.2):
    """Synthetic evaluation workload: model factory (+ oracle factory) and MD frames.  `structure`
    = (lattice, fractional positions, atomic numbers) replaces the rocksalt supercell."""
    from ramannoodle_amd.pmodel import PotGNN
    from ramannoodle_amd.structure import ReferenceStructure

    fn, fe, passes = HPARAMS[hparams]
    lattice, ref, zs = structure if structure is not None else rocksalt(*num_cells)
    rng = np.random.default_rng(seed)
    if structure is None:
        positions = md_frames(rng, lattice, ref, frames, t0=t0)
    else:  # general cell: Cartesian sinusoids mapped through the inverse lattice
        n = ref.shape[0]
        nu, phi = rng.uniform(100.0, 800.0, (1, n, 3)), rng.uniform(0.0, 2 * np.pi, (1, n, 3))
        t = (t0 + np.arange(frames))[:, None, None] * 1.0
        cart = 0.05 * np.cos(2 * np.pi * LIGHT_CM_PER_FS * nu * t + phi)
        x = ref[None] + cart @ np.linalg.inv(lattice)
        positions = x - np.floor(x)
    sym = rng.normal(size=(3, 3))
    mean = (sym + sym.T) + np.diag([30.0, 31.0, 29.0])
    std = np.abs(rng.normal(size=(3, 3)))
    std = (std + std.T) * 0.5 + 0.2
    ref_structure = ReferenceStructure(zs, lattice, ref)
    state = {}

    def build(**kw):
        torch.manual_seed(7)
        model = PotGNN(ref_structure, cutoff, fn, fe, passes, 0.0, 5.0, mean, std, **kw)
        if not state:
            state.update(synthetic_state(model, 7))
        model.load_state_dict(state)
        return model

    def build_oracle():
        from oracle import potgnn_oracle as O  # checker / CPU baseline only
        if not state:
            build()
        edges, trip, tmap = O.build_topology(lattice, ref, zs, cutoff)
        proto = build()
        return O.OracleModel(lattice, np.array(zs), edges, trip, tmap,
                             {k: v.clone() for k, v in state.items()}, proto.gauss_coefficient,
                             fn, fe, passes, mean, std)

    return dict(model=build, oracle=build_oracle, positions=positions, lattice=lattice,
                num_atoms=len(zs), hparams=(fn, fe, passes))


# ----------------------------------------------------------------------------- roofline
CONFIGS = {
    # BASELINE.json configs[2] / configs[1]: cells, frames, seed, scaling
    3: dict(cells=(4, 4, 2), frames=10_000, seed=33, scaling="strong"),
    2: dict(cells=(4, 2, 2), frames=1_000, seed=22, scaling="weak"),
}
PROFILE_ROUNDS = ("r06", "r05", "r04", "r03", "r02", "r01")  # newest first: where committed PMC summaries are looked up


def algorithmic_bytes_edge_block(n, e, fn, fe):
    """B_EB per structure and pass: 4 (N Fn + 2 E Fe)."""
    return 4 * (n * fn + 2 * e * fe)


def edge_block_mfma_flops(e, fn, fe):
    """fp32 MFMA FLOPs of the fused EdgeBlock per structure and pass: the three edge-row
    products W5 edge, W4 edge ([E,Fe] x [Fe,2Fe]) and c2 ([E,Fn] x [Fn,2Fe])."""
    return 2 * e * (2 * fe * 2 * fe + fn * 2 * fe)


def git_blob_hash(path):
    """What `git hash-object` prints for the file: sha1 of "blob <size>\\0" + contents."""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def committed_profile(name, n, e, fn, fe):
    """A PMC summary committed under profiles/rNN/ (taken with rocprofv3 --pmc on this very
    workload: the file records its own N, E, Fn, Fe), newest round first; None if absent.

    A record is only replayed under a fresh timing if the kernel it was taken on is the kernel that just ran: every
    record carries the git blob hash of the kernel's source file at profiling time (tools/make_profile_json.py), and
    a record without a stamp, or whose stamp differs from the file in this tree, is refused (None -> `traffic: null`)."""
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", rnd, name)
        if os.path.exists(path):
            rec = json.load(open(path))
            shape = rec.get("workload_shape", [128, 2304, 64, 64])  # r01 files: config 2
            if list(shape) == [n, e, fn, fe]:
                stamp = rec.get("kernel_source")
                try:
                    current = git_blob_hash(os.path.join(ROOT, stamp["file"])) if stamp else None
                except OSError:
                    current = None
                if not stamp or current != stamp.get("git_blob"):
                    return None  # (the newest record of this shape is stale: an older one is older still)
                rec = dict(rec)
                rec["_path"] = os.path.join("profiles", rnd, name)  # named in the bench line next to what it supplies
                return rec
    return None


def nodeblock_roofline(times, n, e, fn, fe, frames, passes, steps, fused, narrow=False, atom=False):
    """The other scatter-aggregate of a pass: B_NB = 4 (E Fe + 2 N Fn) algorithmic bytes per
    structure and pass over the HIP-event time of the NodeBlock kernel."""
    ms, launches = times.get("node_agg", (0.0, 0))
    if not launches or ms <= 0:
        return None
    per = 4 * (e * fe + 2 * n * fn)
    achieved = per * frames * passes * steps / (ms * 1e-3) / 1e9
    traffic = None
    rec = committed_profile("node_narrow_traffic.json" if narrow else "node_atom_traffic.json" if atom
                            else "node_fused_traffic.json", n, e, fn, fe) if (fused or narrow) else None
    if rec:
        traffic = rec["hbm_bytes_per_structure_pass"] * frames * passes * steps / launches
    return {"kernel": "node_tiled_kernel (NodeBlock scatter-aggregate, projections included: one wave per atom tile, the "
                      "tile's contiguous in-edge rows through LDS 64 at a time, one lane per row, per-atom sums from LDS)" if narrow
            else "node_block_atom_kernel (NodeBlock: 16-atom tiles, MFMA c1 projection with the gate on the accumulators, "
                 "per-atom sums in registers, operand rows through a 4-deep LDS-DMA ring)" if fused and atom
            else "node_block_fused_kernel (NodeBlock: MFMA c1 projection + scatter-aggregate)" if fused
            else "node_agg_kernel (NodeBlock scatter-aggregate; its c1 projection is a separate launch)",
            "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
            "traffic": traffic, "traffic_source": rec["_path"] if rec else None,
            "launches": launches, "avg_launch_ms": ms / launches,
            "algorithmic_bytes_per_structure_pass": per}


def projection_roofline(times, e, fn, fe, frames, passes, steps):
    """The HBM-bound kernel of the unfused pipeline, for comparison: an edge projection reads
    E*Fe and writes E*NOUT floats per structure and pass (its output is an intermediate, so
    these are actual, not 'algorithmic', bytes)."""
    for key, nout, name in (("proj_edge_c3", 4 * fe, "rowgemm_mfma_kernel<64,4,2> (c3 edge projection)"),
                            ("proj_edge_c1", 2 * fn, "rowgemm_mfma_kernel<64,4,1> (c1 edge projection)")):
        ms, launches = times.get(key, (0.0, 0))
        if launches and ms > 0:
            total = 4 * (e * fe + e * nout) * frames * passes * steps
            achieved = total / (ms * 1e-3) / 1e9
            return {"kernel": name, "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                    "frac": achieved / 8000.0, "launches": launches, "avg_launch_ms": ms / launches}
    return None


def measure_case(wl, device, steps, warmup, label):
    """One more workload on this GPU (N = 1 extras of the bench line): whole-pass rate from HBM-resident
    positions plus the two scatter-aggregate kernels' HBM figures, measured as the headline's are."""
    model = wl["model"](device=device)
    n, e = model.num_atoms, model.num_edges
    fn, fe, passes = wl["hparams"]
    frames = len(wl["positions"])
    pos = torch.tensor(wl["positions"], device="cuda")
    out = torch.empty((frames, 3, 3), dtype=torch.float64, device="cuda")
    for _ in range(warmup):
        model.calc_polarizabilities_device(pos, out)
    torch.cuda.synchronize()
    model.set_profiling(1000 + (1 << EDGE_AGG_KERNEL_ID) + (1 << NODE_AGG_KERNEL_ID))
    t0 = time.perf_counter()
    for _ in range(steps):
        model.calc_polarizabilities_device(pos, out)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    times = model.kernel_times()
    model.set_profiling(0)
    flags = model.config_flags()
    fused, narrow = bool(flags["fused_edge_block"]), bool(flags["narrow_kernels"])
    agg_ms, agg_launches = times.get("edge_agg", (0.0, 0))
    per_pass = algorithmic_bytes_edge_block(n, e, fn, fe)
    achieved = per_pass * frames * passes * steps / (agg_ms * 1e-3) / 1e9 if agg_ms > 0 else None
    rec = committed_profile("edge_narrow_traffic.json" if narrow else "edge_fused_traffic.json", n, e, fn, fe) \
        if (fused or narrow) else None
    return {
        "workload": label, "structures_per_s": frames * steps / elapsed, "steps": steps, "frames": frames,
        "num_atoms": n, "num_edges": e, "hparams": {"Fn": fn, "Fe": fe, "P": passes},
        "kernels": "narrow" if narrow else "fused" if fused else "unfused",
        "roofline": {"kernel": "EdgeBlock (projections + triplet scatter-aggregate)", "bound": "hbm",
                     "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0 if achieved else None,
                     "traffic": (rec["hbm_bytes_per_structure_pass"] * frames * passes * steps / agg_launches
                                 if rec and agg_launches else None),
                     "traffic_source": rec["_path"] if rec else None,
                     "launches": agg_launches, "avg_launch_ms": agg_ms / agg_launches if agg_launches else None,
                     "algorithmic_bytes_per_structure_pass": per_pass},
        "roofline_nodeblock": nodeblock_roofline(times, n, e, fn, fe, frames, passes, steps, fused, narrow,
                                                 bool(flags.get("atom_owning_node_block"))),
        # the same frames through the host entry (pageable float64 in, PCIe and the synchronisation inside the clock)
        "host_boundary": host_boundary(model, wl["positions"]),
    }


def tio2_structure():
    """The reference's own TiO2 cell (108 atoms, `test/data/TiO2/POSCAR`) as the committed fixture holds it."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "tio2_notebook.npz"))
    return g["lattice"], g["positions"], [int(z) for z in g["atomic_numbers"]]


def host_pipelined_rate(model, positions, block=2000):
    """The host-buffer boundary at full speed: page-locked blocks through
    `rn_potgnn_calc_polarizabilities_async` (copy-in, evaluation and copy-out of consecutive blocks overlap),
    one warm pass then one timed pass over ALL frames; structures/s including PCIe both ways."""
    from ramannoodle_amd.io._stream import PinnedArray
    frames, atoms = positions.shape[0], positions.shape[1]
    dev = model.device_index
    src, dst = PinnedArray((frames, atoms, 3), dev), PinnedArray((frames, 3, 3), dev)
    try:
        src.array[...] = positions
        rate = None
        for timed in (False, True):
            t0 = time.perf_counter()
            for lo in range(0, frames, block):
                hi = min(lo + block, frames)
                model.calc_polarizabilities_async(src.array[lo:hi], dst.array[lo:hi])
            model.wait()
            if timed:
                rate = frames / (time.perf_counter() - t0)
        return rate
    finally:
        src.free()
        dst.free()


def host_api_rate(model, positions, reps=2):
    """`PotGNN.calc_polarizabilities(numpy array)` -- the call `Trajectory.get_raman_spectrum` makes
    (dynamics/_trajectory.py:71-90): caller-owned pageable float64 in, float64 out, PCIe both ways and the
    synchronisation inside the clock.  One warm call (staging buffers), then the best of `reps`."""
    model.calc_polarizabilities(positions[:max(1, min(len(positions), 64))])
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        model.calc_polarizabilities(positions)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return len(positions) / best


def resident_rate(model, positions, reps=2):
    pos = torch.tensor(positions, device="cuda")
    out = torch.empty((len(positions), 3, 3), dtype=torch.float64, device="cuda")
    model.calc_polarizabilities_device(pos, out, synchronize=True)
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.calc_polarizabilities_device(pos, out, synchronize=True)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return len(positions) / best


def batch1_latency_us(model, positions, calls=300):
    """One structure per call through `calc_polarizabilities`: what the reference's unchanged `Phonons` loop issues
    (dynamics/_phonon.py:93-106: 2 M calls of batch 1).  Microseconds per call, host array in, host array out."""
    for i in range(20):
        model.calc_polarizabilities(positions[i % len(positions)][None])
    t0 = time.perf_counter()
    for i in range(calls):
        model.calc_polarizabilities(positions[i % len(positions)][None])
    return (time.perf_counter() - t0) / calls * 1e6


def host_boundary(model, positions, share=1250):
    """The host boundary where callers use it: the whole trajectory, an 8-GPU share of it, one
    structure at a time -- each next to the HBM-resident rate of the same frames."""
    full_host, full_res = host_api_rate(model, positions), resident_rate(model, positions)
    out = {"entry": "PotGNN.calc_polarizabilities(numpy float64) -> rn_potgnn_calc_polarizabilities: positions cast to "
                    "float32 into page-locked staging, one work chunk at a time (cast / H2D / kernels of consecutive chunks overlapped); "
                    "D2H and sync inside",
           "frames": len(positions), "host_structures_per_s": full_host, "resident_structures_per_s": full_res,
           "host_over_resident": full_host / full_res}
    if len(positions) > share:
        h, r = host_api_rate(model, positions[:share], 3), resident_rate(model, positions[:share], 3)
        out["share_of_8_gpus"] = {"frames": share, "host_structures_per_s": h, "resident_structures_per_s": r,
                                  "host_over_resident": h / r}
    out["batch_1_latency_us_per_call"] = batch1_latency_us(model, positions)
    return out


def cpu_baseline(workload, sample, reps=3):
    """the oracle's faithful restatement of the reference CPU path
    (100-frame sub-batches, N^2 geometry, materialised concat, O(S^2 E) readout) on `sample`
    frames of the same workload, 1 warm-up + `reps` timed repetitions (value = median; min
    next to it).  The O(S E) variant of the same arithmetic and a single-thread run are
    reported beside it so that the GPU/CPU ratio is not inflated by the reference's avoidable
    quadratic terms."""
    from oracle import potgnn_oracle as O
    # a 1-GPU box grants ~16 host cores however many the machine has; more torch threads
    # than that only oversubscribes (measured: 256 threads were 3x slower than 16)
    threads = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), 16)
    model = workload["oracle"]()
    pos = workload["positions"][:sample]

    def timed(frames, faithful, nthreads, repeat):
        torch.set_num_threads(nthreads)
        O.calc_polarizabilities(model, pos[:min(frames, 4)], faithful=faithful)  # warm-up
        out = []
        for _ in range(repeat):
            t0 = time.perf_counter()
            O.calc_polarizabilities(model, pos[:frames], faithful=faithful)
            out.append(time.perf_counter() - t0)
        return out

    faithful = timed(sample, True, threads, reps)
    sane = timed(sample, False, threads, 1)
    few = max(sample // 10, 2)
    single = timed(few, True, 1, 1)
    torch.set_num_threads(threads)
    median = sorted(faithful)[len(faithful) // 2]
    return {"value": sample / median, "unit": "structures/s", "cores": threads, "kind": "port",
            "sample": f"{sample} frames of the same workload (one reference sub-batch), oracle faithful "
                      f"variant (N^2 geometry, materialised concat, O(S^2 E) readout), 1 warm-up + "
                      f"{reps} repetitions of {median:.1f} s (median); deviation from SURVEY.md 8(d), which asked for "
                      f"1000 frames scaled linearly: the faithful variant walks the reference's 100-frame sub-batches, "
                      f"so its cost is linear in the sub-batch count and one sub-batch x (1 + {reps}) runs already "
                      f"takes {median * (1 + reps):.0f} s of the bench's few minutes",
            "repetitions_s": [round(t, 3) for t in faithful],
            "best_structures_per_s": sample / min(faithful),
            "linear_variant_structures_per_s": sample / sane[0],
            "single_thread_structures_per_s": few / single[0]}


# ----------------------------------------------------------------------------- launch
def visible_gpu_count():
    """GPUs this process may use, WITHOUT opening the HIP runtime in it (torch.cuda.device_count() falls
    through to hipGetDeviceCount on ROCm builds without amdsmi): compute nodes of the KFD topology,
    narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None = unknown (the ranks then find out)."""
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(nodes):
        return 0  # no amdgpu compute driver on this machine
    try:
        count = 0
        for node in os.listdir(nodes):
            with open(os.path.join(nodes, node, "properties")) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                count += 1
    except OSError:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        listed = os.environ.get(var)
        if listed is not None:
            count = min(count, len([d for d in listed.split(",") if d.strip() != ""]))
    return count


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes.  The parent
    never touches the GPU (devices are counted from sysfs); the children are fresh interpreters.  The
    rendezvous port is picked by binding port 0 and releasing it, which another process can win before
    rank 0 binds it: a rank that finds the address in use exits with code 98 and the launch is retried on a
    fresh port."""
    import socket
    import subprocess
    share = os.environ.get("RN_BENCH_SHARE_GPU", "0") == "1"
    have = visible_gpu_count()
    if not share and have is not None and have < n:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible")
    for attempt in range(3):
        code = _spawn_once(n, socket, subprocess)
        if code != 98 or attempt == 2:
            return code
    return code


def _spawn_once(n, socket, subprocess):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    codes = [None] * n
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):  # a rank died: the others would wait for ever
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill()
                    codes[i] = p.wait()
            break
        time.sleep(0.05)
    if any(c == 98 for c in codes):
        return 98
    return max((abs(c) for c in codes), default=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=3,
                    help="BASELINE.json workload: 3 = 256 atoms, 10k frames sharded over the GPUs "
                         "(default); 2 = 128 atoms, 1000 frames per GPU")
    ap.add_argument("--frames", type=int, default=0,
                    help="override: total frames (config 3) / frames per GPU (config 2)")
    ap.add_argument("--cells", type=str, default="")
    ap.add_argument("--hparams", choices=list(HPARAMS), default="perf")
    ap.add_argument("--cpu-sample", type=int, default=100,
                    help="frames for the CPU baseline (100 = one reference sub-batch)")
    ap.add_argument("--cpu-reps", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the N = 1 extras: pipelined host-buffer rate, exact-fp32 comparison run, documented "
                         "widths, the reference's published TiO2 workload (profiling passes: the timed steps only)")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="time the steps without the HIP-event pairs around the EdgeBlock / NodeBlock launches "
                         "(the A/B that prices them: profiles/r04/bench_event_overhead.txt); roofline is then empty")
    ap.add_argument("--profile-all", action="store_true",
                    help="HIP-event timing of every kernel (perturbs the timed region)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    # (test hook: RN_BENCH_SHARE_GPU=1 runs every rank on cuda:0 over gloo, so that the N > 1
    #  code path can be exercised on a 1-GPU box; RCCL refuses two ranks on one device)
    share_gpu = os.environ.get("RN_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811
        try:
            if share_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        except Exception as exc:  # pylint: disable=broad-except
            if "address already in use" in str(exc).lower() or "eaddrinuse" in str(exc).lower():
                sys.exit(98)  # spawn_ranks retries on a fresh port
            raise

    from ramannoodle_amd.parallel import shard_bounds
    cfg = CONFIGS[args.config]
    cells = tuple(int(c) for c in args.cells.split(",")) if args.cells else cfg["cells"]
    strong = cfg["scaling"] == "strong"
    if strong:  # ONE trajectory of `total` frames; this rank owns the contiguous block [lo, hi)
        total = args.frames or cfg["frames"]
        lo, hi, per = shard_bounds(total, world, rank)
    else:       # every rank has its own `frames` frames (consecutive stretches of one trajectory)
        per = args.frames or cfg["frames"]
        total, lo, hi = per * world, rank * per, (rank + 1) * per
    mine = hi - lo
    wl = make_workload(cells, mine, args.hparams, seed=cfg["seed"], t0=lo)
    model = wl["model"](device=local, max_chunk_structures=args.chunk)
    n, e = model.num_atoms, model.num_edges
    fn, fe, passes = wl["hparams"]
    pos = torch.tensor(wl["positions"], device="cuda")
    # the rank's block of the all-gather input (padded to `per` frames when the split is ragged)
    out = torch.zeros((per, 3, 3), dtype=torch.float64, device="cuda")
    gathered = torch.empty((world * per, 3, 3), dtype=torch.float64, device="cuda")

    def step():
        if mine:
            model.calc_polarizabilities_device(pos, out[:mine])
        if world > 1:
            dist.all_gather_into_tensor(gathered, out)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if not args.no_kernel_events:
        model.set_profiling(1 if args.profile_all
                            else 1000 + (1 << EDGE_AGG_KERNEL_ID) + (1 << PROJ_C3_KERNEL_ID) + (1 << NODE_AGG_KERNEL_ID))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    times = model.kernel_times()
    model.set_profiling(0)
    # informational, N > 1: the same step from the caller's HOST array (what `calc_polarizabilities_sharded` does: every rank
    # uploads its block through the pipelined staging entry, results stay in HBM for the all-gather) -- the rate a
    # `Trajectory.get_raman_spectrum` across ranks sees, beside the HBM-resident `value`
    host_inclusive = None
    if world > 1 and not args.no_extras:
        host_steps = max(1, min(args.steps, 5))

        def host_step():
            if mine:
                model.calc_polarizabilities_to_device(wl["positions"], out[:mine])
            dist.all_gather_into_tensor(gathered, out)

        host_step()
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(host_steps):
            host_step()
        torch.cuda.synchronize()
        dist.barrier()
        th = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
        dist.all_reduce(th, op=dist.ReduceOp.MAX)
        host_inclusive = total * host_steps / float(th.item())
    # informational: the host-buffer entry point (adds PCIe H2D/D2H); never the headline value
    host_rate = None
    host_api = None
    if rank == 0 and mine and not args.no_extras:
        host_rate = host_pipelined_rate(model, wl["positions"])
        if world == 1:
            host_api = host_boundary(model, wl["positions"])

    # informational (N = 1): the same step with the matrix products on the exact-fp32 MFMA instead of the
    # split-f16 products, and how far the two outputs are apart -- so that the headline can be read
    # against a run that makes no use of f16 at all
    exact = None
    if world == 1 and mine and model.config_flags()["split_f16_mfma"] and model.config_flags()["fused_edge_block"] \
            and "RN_POTGNN_MFMA" not in os.environ and not args.no_extras:
        os.environ["RN_POTGNN_MFMA"] = "f32"
        try:
            model32 = wl["model"](device=local, max_chunk_structures=args.chunk)
            out32 = torch.zeros_like(out)
            model32.calc_polarizabilities_device(pos, out32[:mine])
            torch.cuda.synchronize()
            model32.set_profiling(1000 + (1 << EDGE_AGG_KERNEL_ID) + (1 << NODE_AGG_KERNEL_ID))
            t1 = time.perf_counter()
            for _ in range(2):
                model32.calc_polarizabilities_device(pos, out32[:mine])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / 2
            times32 = model32.kernel_times()
            model32.set_profiling(0)
            model.calc_polarizabilities_device(pos, out[:mine])
            torch.cuda.synchronize()
            diff = float((out[:mine] - out32[:mine]).abs().max() / out32[:mine].abs().max())
            ms32, launches32 = times32.get("edge_agg", (0.0, 0))
            bytes32 = algorithmic_bytes_edge_block(n, e, fn, fe) * mine * passes * 2
            exact = {"structures_per_s": mine / dt, "max_rel_diff_of_alpha": diff,
                     "note": "RN_POTGNN_MFMA=f32: v_mfma_f32_16x16x4_f32 everywhere, no f16 operands",
                     # the dominant kernel of THIS leg, HIP-event timed like the headline's (its F16 = false instantiation)
                     "roofline_exact_fp32": ({"kernel": "edge_block_ps_kernel<F16 = false> (f32 weight fragments and operand "
                                                        "tiles, v_mfma_f32_16x16x4_f32)",
                                              "bound": "hbm", "achieved": bytes32 / (ms32 * 1e-3) / 1e9, "peak": 8000.0,
                                              "unit": "GB/s", "frac": bytes32 / (ms32 * 1e-3) / 1e9 / 8000.0, "traffic": None,
                                              "launches": launches32, "avg_launch_ms": ms32 / launches32,
                                              "mfma_frac": edge_block_mfma_flops(e, fn, fe) * mine * passes * 2
                                              / (ms32 * 1e-3) / 157.3e12}
                                             if launches32 and ms32 > 0 else None)}
            del model32
        finally:
            del os.environ["RN_POTGNN_MFMA"]

    if rank == 0:
        agg_ms, agg_launches = times.get("edge_agg", (0.0, 0))
        flags = model.config_flags()
        fused, narrow = bool(flags["fused_edge_block"]), bool(flags["narrow_kernels"])
        per_pass = algorithmic_bytes_edge_block(n, e, fn, fe)
        total_bytes = per_pass * mine * passes * args.steps
        achieved = total_bytes / (agg_ms * 1e-3) / 1e9 if agg_ms > 0 else None
        peak = 8000.0
        role_split = bool(fused and flags.get("role_split_edge_block"))
        traffic_rec = committed_profile("edge_narrow_traffic.json" if narrow else "edge_ps_traffic.json" if role_split
                                        else "edge_fused_traffic.json" if fused else "edge_agg_traffic.json", n, e, fn, fe)
        issue_rec = committed_profile("edge_narrow_issue.json" if narrow else "edge_ps_issue.json" if role_split
                                      else "edge_fused_issue.json", n, e, fn, fe) if (fused or narrow) else None
        frames_per_launch = mine * passes * args.steps / agg_launches if agg_launches else None
        roofline = {
            "kernel": ("edge_narrow_kernel (EdgeBlock: projections + triplet scatter-aggregate, one lane per "
                       "destination edge, two columns per packed-f32 instruction)" if narrow
                       else "edge_block_ps_kernel (EdgeBlock: MFMA projections by producer waves + triplet "
                            "scatter-aggregate by consumer waves, one 12-wave workgroup per CU)"
                       if fused and flags.get("role_split_edge_block")
                       else "edge_block_fused_kernel (EdgeBlock: MFMA projections + triplet scatter-aggregate)"
                       if fused else "edge_agg_kernel (EdgeBlock triplet scatter-aggregate)"),
            "bound": "hbm",
            "achieved": achieved,
            "peak": peak,
            "unit": "GB/s",
            "frac": achieved / peak if achieved else None,
            # traffic and issue_frac are NOT measured in this run: they replay the committed rocprofv3 PMC / SQ passes of
            # this kernel on this workload shape (the files named in *_source; None when no record matches)
            "traffic": (traffic_rec["hbm_bytes_per_structure_pass"] * frames_per_launch
                        if traffic_rec and frames_per_launch else None),
            "traffic_source": traffic_rec["_path"] if traffic_rec else None,
            "launches": agg_launches,
            "avg_launch_ms": agg_ms / agg_launches if agg_launches else None,
            "algorithmic_bytes_per_structure_pass": per_pass,
            # why frac is small: the kernel is bound by SIMD instruction issue, not by HBM.
            # issue_frac = (VALU-busy + fp32-MFMA-busy) share of SIMD cycles from the committed SQ
            # counter pass of this kernel on this workload; mfma_frac = its fp32 MFMA FLOP/s, measured
            # live, over the 157.3 TFLOP/s fp32 peak.
            "issue_frac": issue_rec["valu_plus_mfma_busy"] if issue_rec else None,
            "issue_source": issue_rec["_path"] if issue_rec else None,
            # matrix work of the kernel (three [E,64]x[64,128] products per pass, as 3 split-f16 products
            # each on the f16 MFMA) in fp32-equivalent FLOP/s over the 157.3 TFLOP/s fp32 peak
            "mfma_frac": (edge_block_mfma_flops(e, fn, fe) * mine * passes * args.steps / (agg_ms * 1e-3) / 157.3e12
                          if fused and not narrow and agg_ms > 0 else None),
            "note": "the EdgeBlock is SIMD-issue bound (DESIGN.md section 5): about 10 VALU instructions, "
                    "3 of them transcendental, per (triplet, gate); achieved/peak/frac are the HBM "
                    "figures the metric asks for, issue_frac (VALU-busy + MFMA-busy share of SIMD cycles, "
                    "from the committed SQ counter pass) says what actually bounds the kernel; "
                    "roofline_nodeblock is the pass's other scatter-aggregate, the one that streams",
        }
        result = {
            "metric": "structures/sec (GNN polarizability eval)",
            "value": total * args.steps / elapsed,
            "unit": "structures/s",
            "n_gpus": world,
            # what torch.distributed reported after init_process_group("nccl") (= RCCL on ROCm); 1 without a group
            "rccl_ranks_seen": (dist.get_world_size() if world > 1 else 1),
            "collective_backend": (dist.get_backend() if world > 1 else None),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": cfg["scaling"],
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config {args.config}: PotGNN eval, rocksalt {cells[0]}x{cells[1]}x{cells[2]} "
                            f"({n} atoms, E={e}, cutoff 3.2 A), "
                            + (f"one {total}-frame MD trajectory sharded over {world} GPU(s)" if strong
                               else f"{per} MD frames per GPU")
                            + f", Fn={fn} Fe={fe} P={passes}",
                "total_frames": total,
                "frames_per_gpu": per,
                "hparams": args.hparams,
                "parallelism": (f"frames sharded x{world} in contiguous blocks "
                                f"({'strong' if strong else 'weak'} scaling), one RCCL all-gather of alpha per step"
                                if world > 1 else "single GPU"),
                # how the fp32 matrix products of the pass are evaluated
                "matrix_products": ("fp32 operands split exactly into two f16 halves (22 significant bits), three "
                                    "f16 MFMA products with fp32 accumulation; measured error vs float64 6e-7, as "
                                    "exact-fp32 MFMA (RN_POTGNN_MFMA=f32)" if flags["split_f16_mfma"] and fused
                                    else "fp32 scalar FMA" if narrow else "exact fp32 MFMA"),
                # layout of the edge embedding in HBM between the kernels of a pass (include/rn_potgnn.h, flag bit 10)
                "edge_rows": ("split-f16 operand pairs (f16 hi x8 | f16 lo x8 per eight columns, 256 B per row)"
                              if flags.get("split_f16_pair_rows") else "float32"),
            },
            "roofline": roofline,
            "roofline_nodeblock": nodeblock_roofline(times, n, e, fn, fe, mine, passes, args.steps, fused, narrow,
                                                     bool(flags.get("atom_owning_node_block"))),
            "roofline_projection": projection_roofline(times, e, fn, fe, mine, passes, args.steps),
            # the same step through the C ABI's host-buffer entry, pipelined (PCIe both ways inside the clock);
            # `value` is the HBM-resident figure the contract asks for
            "host_pipelined_structures_per_s": host_rate,
            "host_over_resident": host_rate / (total * args.steps / elapsed) if host_rate and world == 1 else None,
            # what callers of the Python API get (pageable float64 arrays): whole trajectory, an 8-GPU share of it,
            # one structure per call
            "host_boundary": host_api,
            # N > 1: `value` from the caller's host array instead of HBM-resident positions (upload inside the clock)
            "host_inclusive_structures_per_s": host_inclusive,
            "exact_fp32_mfma": exact,
        }
        if world == 1 and not args.no_extras and args.config == 3 and args.hparams == "perf" and not args.frames \
                and not args.cells:
            # (b) the documented widths on the same cell and trajectory, and the one workload the reference
            # publishes a rate for (506.79 configs/s, machine-learning.ipynb:401)
            del model
            torch.cuda.empty_cache()
            doc = make_workload(cells, mine, "parity", seed=cfg["seed"], t0=lo)
            result["documented_hparams"] = measure_case(
                doc, local, 3, 1, f"same cell and trajectory as `config`, the reference's documented widths "
                                  f"Fn=5 Fe=14 P=4 (machine-learning.ipynb:187-192)")
            del doc
            tio2 = make_workload(frames=20_000, hparams="parity", seed=108, structure=tio2_structure(), cutoff=2.0)
            pub = measure_case(tio2, local, 3, 1, "the reference's published workload: TiO2 108 atoms, cutoff 2 A, "
                                                  "Fn=5 Fe=14 P=4, 20000 MD frames (synthetic frames and weights)")
            pub["published_configs_per_s"] = 506.79
            pub["vs_published"] = pub["structures_per_s"] / 506.79
            pub["published_source"] = "docs/source/notebooks/machine-learning.ipynb:401 (hardware unstated)"
            result["reference_published_workload"] = pub
        if args.profile_all:
            result["kernel_ms"] = {k: round(v[0], 3) for k, v in times.items()}
        if world == 1 and not args.no_cpu:
            result["cpu_baseline"] = cpu_baseline(wl, min(args.cpu_sample, mine), args.cpu_reps)
        for key in ("roofline_nodeblock", "roofline_projection"):
            if result.get(key) is None:
                result.pop(key, None)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
