"""Frame sharding + the single all-gather, exercised with 2 CPU processes over gloo
(the N > 1 path of bench.py / ramannoodle_amd.parallel; on the GPU box the same code runs
over RCCL).  A deterministic host-side stand-in model replaces the device model here so
the test needs no GPU -- it checks the sharding/collective logic, not the kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ramannoodle_amd.parallel import (all_gather_frames, calc_polarizabilities_sharded,
                                      calc_raman_tensors_sharded, shard_bounds)


class _Model:
    def __init__(self, n):
        self.w = np.random.default_rng(5).normal(size=(n * 3, 6))
        self.calls = []

    def calc_polarizabilities(self, positions_batch):
        self.calls.append(positions_batch.shape[0])
        v = positions_batch.reshape(positions_batch.shape[0], self.w.shape[0]) @ self.w
        return v[:, [[0, 3, 4], [3, 1, 5], [4, 5, 2]]]

    def calc_raman_tensors(self, ref_positions, displacements, delta=1e-3):
        plus = self.calc_polarizabilities(ref_positions[None] + delta * displacements)
        minus = self.calc_polarizabilities(ref_positions[None] - delta * displacements)
        return (plus - minus) / delta


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 7
        pos = np.random.default_rng(9).uniform(size=(total, n, 3))
        model = _Model(n)
        full = calc_polarizabilities_sharded(model, pos)
        lo, hi, per = shard_bounds(total, world, rank)
        assert model.calls == [hi - lo]
        np.save(os.path.join(out_dir, f"r{rank}.npy"), full)
        # the bare collective on a ragged split
        local = torch.arange(lo, hi, dtype=torch.float64).view(-1, 1, 1).expand(-1, 3, 3).contiguous()
        got = all_gather_frames(local, total)
        assert torch.equal(got[:, 0, 0], torch.arange(total, dtype=torch.float64))
        # phonon modes shard the same way
        disp = np.random.default_rng(3).normal(size=(total, n, 3))
        tensors = calc_raman_tensors_sharded(_Model(n), pos[0], disp, delta=1e-2)
        np.save(os.path.join(out_dir, f"t{rank}.npy"), tensors)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world, total", [(2, 11), (2, 8), (2, 1), (3, 11), (3, 4)])
def test_sharded_evaluation_ranks(tmp_path, world, total):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    n = 7
    pos = np.random.default_rng(9).uniform(size=(total, n, 3))
    expect = _Model(n).calc_polarizabilities(pos)
    disp = np.random.default_rng(3).normal(size=(total, n, 3))
    expect_tensors = _Model(n).calc_raman_tensors(pos[0], disp, delta=1e-2)
    for r in range(world):
        # (the stand-in's BLAS product may round differently for a 1-row block: 1e-15)
        np.testing.assert_allclose(np.load(tmp_path / f"r{r}.npy"), expect, rtol=1e-13, atol=1e-13)
        np.testing.assert_allclose(np.load(tmp_path / f"t{r}.npy"), expect_tensors, rtol=1e-12, atol=1e-12)


def _gather8_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for total in (10_000, 1_536, 10_003):  # config 3's frames, config 4's displaced cells, a ragged split
            lo, hi, per = shard_bounds(total, world, rank)
            local = (torch.arange(lo, hi, dtype=torch.float64).view(-1, 1, 1) * torch.ones(1, 3, 3, dtype=torch.float64))
            got = all_gather_frames(local.contiguous(), total)
            assert got.shape == (total, 3, 3)
            assert torch.equal(got[:, 1, 2], torch.arange(total, dtype=torch.float64))
            if rank == 0:
                np.save(os.path.join(out_dir, f"sizes_{total}.npy"), np.array([lo, hi, per]))
    finally:
        dist.destroy_process_group()


def test_eight_rank_partition_of_the_baseline_configs(tmp_path):
    """The exact N = 8 partitions the scaling run uses (BASELINE configs 3 and 4): 10 000 frames ->
    8 x 1250, 1536 displaced cells -> 8 x 192 (= 96 modes x +-), plus a ragged 10 003; the padded
    all-gather returns every item once, in order, on every rank."""
    mp.spawn(_gather8_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    assert np.load(tmp_path / "sizes_10000.npy").tolist() == [0, 1250, 1250]
    assert np.load(tmp_path / "sizes_1536.npy").tolist() == [0, 192, 192]
    assert np.load(tmp_path / "sizes_10003.npy").tolist() == [0, 1251, 1251]
    assert [shard_bounds(10_000, 8, r)[:2] for r in range(8)] == [(1250 * r, 1250 * (r + 1)) for r in range(8)]


class _AnalyticModel(_Model):
    """Stand-in with the device model's analytic entry: d(alpha)/d(r) contracted with the modes (`method="analytic"`)."""

    def _check_raman_arguments(self, ref_positions, displacements, **kwargs):
        if kwargs.get("method", "finite-difference") not in ("finite-difference", "analytic"):
            raise ValueError("method")
        if displacements.shape[1:] != ref_positions.shape:
            raise ValueError("displacements has wrong shape")

    def calc_raman_tensors(self, ref_positions, displacements, delta=1e-3, method="finite-difference"):
        if method != "analytic":
            return super().calc_raman_tensors(ref_positions, displacements, delta)
        self.calls.append(("analytic", displacements.shape[0]))
        v = displacements.reshape(displacements.shape[0], self.w.shape[0]) @ self.w * 2.0  # (plus - minus) / delta of a linear model
        return v[:, [[0, 3, 4], [3, 1, 5], [4, 5, 2]]]


def _analytic8_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, modes = 256, 768  # BASELINE config 4: 256 atoms, 768 modes
        rng = np.random.default_rng(44)
        ref = rng.uniform(size=(n, 3))
        disp = rng.normal(size=(modes, n, 3))
        model = _AnalyticModel(n)
        tensors = calc_raman_tensors_sharded(model, ref, disp, method="analytic")
        assert model.calls == [("analytic", 96)], model.calls  # this rank's block only: 768 modes / 8 ranks
        np.save(os.path.join(out_dir, f"a{rank}.npy"), tensors)
        # a bad argument raises on every rank before any collective
        try:
            calc_raman_tensors_sharded(model, ref, disp[:, :-1], method="analytic")
        except ValueError:
            np.save(os.path.join(out_dir, f"raised{rank}.npy"), np.ones(1))
    finally:
        dist.destroy_process_group()


def test_eight_rank_analytic_raman_tensors_of_config4(tmp_path):
    """`calc_raman_tensors_sharded(method="analytic")` at the N = 8 partition of BASELINE config 4 (768 modes of a 256-atom
    cell -> 96 modes per rank, one all-gather of float64[96,3,3] blocks): every rank evaluates its block only and returns
    all 768 tensors, equal to the single-process result."""
    mp.spawn(_analytic8_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    rng = np.random.default_rng(44)
    ref = rng.uniform(size=(256, 3))
    disp = rng.normal(size=(768, 256, 3))
    expect = _AnalyticModel(256).calc_raman_tensors(ref, disp, method="analytic")
    for r in range(8):
        got = np.load(tmp_path / f"a{r}.npy")
        assert got.shape == (768, 3, 3)
        np.testing.assert_allclose(got, expect, rtol=1e-12, atol=1e-12)
        assert (tmp_path / f"raised{r}.npy").exists()


def test_shard_bounds_cover_everything():
    for total in (0, 1, 7, 8, 10000):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for (lo, hi, per), (lo2, _, _) in zip(spans, spans[1:]):
                assert hi == lo2 and hi - lo <= per
            sizes = [hi - lo for lo, hi, _ in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == spans[0][2]
            assert total < world or min(sizes) >= 1  # no empty rank while there is work for all


class _TorchModel(torch.nn.Module):
    """Host stand-in with the surface ``train_single_epoch`` uses; BatchNorm-free, so the
    data-parallel step must equal the single-process step on the whole mini-batch."""

    def __init__(self, group=None):
        super().__init__()
        torch.manual_seed(3)
        self.lin = torch.nn.Linear(6, 6)
        self.data_parallel_group = group

    def forward(self, lattice, atomic_numbers, position):  # pylint: disable=unused-argument
        return self.lin(position.reshape(position.shape[0], -1)[:, :6])


def _dataset(n):
    rng = np.random.default_rng(1)
    pos = torch.tensor(rng.normal(size=(n, 2, 3)), dtype=torch.float32)
    target = torch.tensor(rng.normal(size=(n, 6)), dtype=torch.float32)
    lat = torch.eye(3).expand(n, 3, 3)
    zs = torch.ones(n, 2, dtype=torch.int32)
    return torch.utils.data.TensorDataset(lat, zs, pos, target)


def _train_worker(rank, world, port, out_dir, items=16):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ramannoodle_amd.pmodel.train import train_single_epoch
        model = _TorchModel(group=dist.group.WORLD)
        opt = torch.optim.SGD(model.parameters(), lr=0.1)
        losses = train_single_epoch(model, _dataset(items), _dataset(4), 8, opt, torch.nn.MSELoss())
        torch.save({"state": model.state_dict(), "losses": losses}, os.path.join(out_dir, f"dp{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world, items", [(2, 16), (3, 19), (3, 16)])
def test_data_parallel_epoch_equals_single_process(tmp_path, world, items):
    """``train_single_epoch`` with a process group: same shuffled mini-batches on every rank,
    contiguous balanced blocks, rank losses weighted by block size, gradients averaged -> same
    parameters and training loss as one process on full batches.  Three ranks with batches of
    8 (blocks 3, 3, 2) and a last batch of 3 (19 items) are the ragged cases a ceil-division
    split left a rank empty on."""
    from ramannoodle_amd.pmodel.train import train_single_epoch
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path), items), nprocs=world, join=True)
    single = _TorchModel()
    opt = torch.optim.SGD(single.parameters(), lr=0.1)
    want = train_single_epoch(single, _dataset(items), _dataset(4), 8, opt, torch.nn.MSELoss())
    for r in range(world):
        got = torch.load(tmp_path / f"dp{r}.pt", weights_only=False)
        for k, v in single.state_dict().items():
            torch.testing.assert_close(got["state"][k], v, rtol=1e-5, atol=1e-6)
        assert got["losses"][0] == pytest.approx(want[0], rel=1e-5)
        assert got["losses"][1] == pytest.approx(want[1], rel=1e-5)


class _CheckedModel(_Model):
    """Stand-in with the argument checks the device model runs (``PotGNN._check_raman_arguments``)."""

    def _check_raman_arguments(self, ref_positions, displacements, delta=1e-3, method="finite-difference", **unknown):
        if unknown:
            raise TypeError(f"unexpected keyword arguments: {sorted(unknown)}")
        if displacements.ndim != 3 or displacements.shape[1:] != ref_positions.shape:
            raise ValueError("displacements has wrong shape")
        if delta == 0:
            raise ValueError("delta must be non-zero")


def _bad_argument_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, total = 5, 1  # ONE mode over two ranks: rank 1 owns an empty block
        pos = np.random.default_rng(1).uniform(size=(n, 3))
        disp = np.random.default_rng(2).normal(size=(total, n, 3))
        raised = []
        for kwargs, bad_disp in (({"delta": 0.0}, disp), ({"delat": 1e-3}, disp), ({}, disp[:, :-1])):
            try:
                calc_raman_tensors_sharded(_CheckedModel(n), pos, bad_disp, **kwargs)
                raised.append("none")
            except (ValueError, TypeError) as exc:
                raised.append(type(exc).__name__)
        # and a good call still works afterwards (nobody is stuck in a collective)
        good = calc_raman_tensors_sharded(_CheckedModel(n), pos, disp, delta=1e-2)
        assert good.shape == (total, 3, 3)
        with open(os.path.join(out_dir, f"bad{rank}.txt"), "w") as f:
            f.write(",".join(raised))
    finally:
        dist.destroy_process_group()


def test_bad_arguments_raise_on_every_rank(tmp_path):
    """the sharded phonon entry validates on EVERY rank before it branches on the rank's block, so a bad
    ``delta``, shape or keyword raises everywhere -- also on a rank whose block is empty -- instead of leaving the
    other ranks waiting in the all-gather."""
    port = _free_port()
    mp.spawn(_bad_argument_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert (tmp_path / f"bad{r}.txt").read_text() == "ValueError,TypeError,ValueError"
