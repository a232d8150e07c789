"""CPU oracle for the PotGNN polarizability-evaluation path.  TEST INFRASTRUCTURE ONLY.

A plain-torch (CPU) restatement of the reference algorithm, function by function, with
the reference ``file:line`` each follows (paths relative to ``/root/reference``).  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module; the product (``ramannoodle_amd``) never does.

Pinning: checked against the golden fixtures in ``tests/golden/*.npz`` that were
produced by executing the reference itself (``tests/golden/make_golden.py``): graph and
triplet indices bit-exact, every per-stage intermediate and the final polarizabilities
to fp32 round-off.  One caveat: the triplet *ordering* comes from
``torch_geometric.nn.models.dimenet.triplets``, a third-party dependency absent from
this image (pinned by the reference only as ``torch_geometric >= 2.3.0``,
``pyproject.toml:46``).  Its published algorithm is restated in :func:`triplets`; the
fixtures used the same restatement, so parity is UNPINNED at that ordering (it only
affects fp32 summation order, not the triplet set).

Two evaluation variants are provided:

* ``forward(..., faithful=True)``  -- the reference's data flow: ``[S,N,N,3]`` pairwise
  geometry, materialised triplet concatenation, O(S^2 E) per-structure readout loop.
* ``forward(..., faithful=False)`` -- same arithmetic per element but O(S E) data flow
  (edge-list geometry, segment mean).  Used as the "sane" CPU baseline.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

LOG2 = math.log(2.0)

# 6-vector <-> 3x3 index tables (ramannoodle/dataset/torch/utils.py:30-37, 56-57)
VEC_TO_TENSOR = [[0, 3, 4], [3, 1, 5], [4, 5, 2]]
TENSOR_TO_VEC = ([0, 1, 2, 0, 0, 1], [0, 1, 2, 1, 2, 2])


# ----------------------------------------------------------------------------- graph
def min_image_cart(lattice: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
    """All-pairs minimum-image Cartesian displacement, ``[S,N,N,3]``.

    ``out[s,a,b] = wrap(x_b - x_a) @ lattice`` with ``wrap(d) = d%1 - 1 if d%1 > 0.5
    else d%1`` (ramannoodle/pmodel/torch/_gnn.py:603-610, _utils.py:122-127).
    """
    d = positions.unsqueeze(1) - positions.unsqueeze(2)
    m = d % 1
    d = torch.where(m > 0.5, m - 1, m)
    return d.matmul(lattice[:, None, :, :].expand(-1, d.size(1), -1, -1))


def radius_graph(lattice: torch.Tensor, positions: torch.Tensor, cutoff: float) -> torch.Tensor:
    """One-time neighbour list ``int64[3,E]`` = (graph, a, b), sorted by (a, b).

    ``dist <= cutoff`` and ``a != b`` on the ``[1,N,N]`` distance matrix, then
    ``nonzero`` (ramannoodle/pmodel/torch/_utils.py:118-137).
    """
    cart = min_image_cart(lattice, positions)
    dist = torch.sqrt(torch.sum(cart**2, dim=-1))
    adj = dist <= cutoff
    n = adj.size(-1)
    adj = torch.logical_and(adj, ~torch.eye(n, dtype=torch.bool).expand(adj.size(0), -1, -1))
    return torch.nonzero(adj).T


def triplets(edge_index: torch.Tensor, num_nodes: int):
    """Edge triplets k->j->i, restating PyG ``dimenet.triplets`` (third-party, absent).

    Called as in ramannoodle/pmodel/torch/_utils.py:157-160 with
    ``edge_index = ref_edge_indexes[[1, 2]]`` so ``row = a``, ``col = b``.  Returns the
    7-tuple in PyG's order ``(col, row, idx_i, idx_j, idx_k, idx_kj, idx_ji)``; the
    reference consumes it *positionally* (_gnn.py:647-650), so tuple slot 5 (PyG's
    ``idx_kj``) is what ``_EdgeBlock`` uses both as its 4th concat operand and as its
    scatter destination, and slot 6 (``idx_ji``) is its 5th concat operand.
    """
    row, col = edge_index[0], edge_index[1]
    e = row.numel()
    key = col * num_nodes + row  # sparse (col,row) ordering of SparseTensor(row=col, col=row)
    order = torch.argsort(key, stable=True)
    s_col, s_val = row[order], order
    ptr = torch.zeros(num_nodes + 1, dtype=torch.long)
    ptr[1:] = torch.cumsum(torch.bincount(col, minlength=num_nodes), 0)
    idx_i, idx_j, idx_k, idx_kj, idx_ji = [], [], [], [], []
    for edge in range(e):  # adj_t[row]: sparse row of node row[edge], for each edge in order
        j, i = int(row[edge]), int(col[edge])
        for p in range(int(ptr[j]), int(ptr[j + 1])):
            k = int(s_col[p])
            if k == i:
                continue
            idx_i.append(i)
            idx_j.append(j)
            idx_k.append(k)
            idx_kj.append(int(s_val[p]))
            idx_ji.append(edge)
    t = lambda v: torch.tensor(v, dtype=torch.long)  # noqa: E731
    return col, row, t(idx_i), t(idx_j), t(idx_k), t(idx_kj), t(idx_ji)


# ----------------------------------------------------------------------------- model
@dataclass
class OracleModel:
    """Weights + frozen topology of one PotGNN (ramannoodle/pmodel/torch/_gnn.py:484-539)."""

    lattice: np.ndarray  # [3,3] f64
    atomic_numbers: np.ndarray  # [N] int
    edges: torch.Tensor  # int64 [3,E]
    trip: tuple  # 7 int64 tensors
    atom_type_map: torch.Tensor  # int [119]
    sd: dict  # state dict (torch tensors)
    coefficient: float
    fn: int
    fe: int
    passes: int
    mean: np.ndarray
    std: np.ndarray
    dtype: torch.dtype = torch.float32
    _trip_cache: dict = field(default_factory=dict)

    @property
    def num_atoms(self) -> int:
        return int(self.atomic_numbers.shape[0])

    @property
    def num_edges(self) -> int:
        return int(self.edges.shape[1])

    @property
    def num_triplets(self) -> int:
        return int(self.trip[2].numel())

    def to(self, dtype: torch.dtype) -> "OracleModel":
        sd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in self.sd.items()}
        return OracleModel(self.lattice, self.atomic_numbers, self.edges, self.trip,
                           self.atom_type_map, sd, self.coefficient, self.fn, self.fe,
                           self.passes, self.mean, self.std, dtype)


def model_from_arrays(arrs: dict, dtype=torch.float32) -> OracleModel:
    """Build from a golden ``.npz`` (or the same keys produced by the product's exporter)."""
    sd = {}
    for k in arrs:
        if k.startswith("sd/"):
            t = torch.from_numpy(np.array(arrs[k]))
            sd[k[3:]] = t.to(dtype) if t.is_floating_point() else t
    hp = arrs["hp"]
    trip = tuple(
        torch.from_numpy(np.array(arrs["trip/" + n])).long()
        for n in ["i", "j", "idx_i", "idx_j", "idx_k", "slot5", "slot6"]
    )
    return OracleModel(
        lattice=np.array(arrs["lattice"], dtype=np.float64),
        atomic_numbers=np.array(arrs["atomic_numbers"]),
        edges=torch.from_numpy(np.array(arrs["ref_edge_indexes"])).long(),
        trip=trip,
        atom_type_map=torch.from_numpy(np.array(arrs["atom_type_map"])),
        sd=sd,
        coefficient=float(arrs["gauss_coefficient"]),
        fn=int(hp[1]), fe=int(hp[2]), passes=int(hp[3]),
        mean=np.array(arrs["mean"]), std=np.array(arrs["std"]), dtype=dtype,
    )


def build_topology(lattice, positions, atomic_numbers, cutoff, dtype=torch.float32):
    """Graph + triplets + atom-type map exactly as ``PotGNN.__init__`` derives them
    (ramannoodle/pmodel/torch/_gnn.py:489-506)."""
    lat = torch.from_numpy(np.asarray(lattice)).unsqueeze(0).type(dtype)
    pos = torch.from_numpy(np.asarray(positions)).unsqueeze(0).type(dtype)
    edges = radius_graph(lat, pos, cutoff)
    trip = triplets(edges[[1, 2]], len(atomic_numbers))
    type_map = (torch.zeros(119) - 1).type(torch.int)
    for t, z in enumerate(set(int(z) for z in atomic_numbers)):
        type_map[z] = t
    return edges, trip, type_map


# ----------------------------------------------------------------------------- ops
def ssp(x):
    """ShiftedSoftplus (PyG schnet): softplus(x) - log 2."""
    return F.softplus(x) - LOG2


def lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def lnorm(sd, name, x):
    w = sd[name + ".weight"]
    return F.layer_norm(x, (w.numel(),), w, sd[name + ".bias"], 1e-5)


def gate(x):
    """sigmoid(first half) * tanh(second half) (ramannoodle/pmodel/torch/_gnn.py:143-145)."""
    f, c = x.chunk(2, dim=1)
    return f.sigmoid() * c.tanh()


def seg_sum(src, index, size):
    """``scatter(..., reduce="sum")`` (PyG) == zero-init index_add."""
    return torch.zeros((size, src.size(1)), dtype=src.dtype).index_add_(0, index, src)


def rotations_ex_to(u: torch.Tensor) -> torch.Tensor:
    """Rodrigues rotation taking e_x onto each row of ``u``
    (ramannoodle/pmodel/torch/_utils.py:24-65)."""
    b = u / torch.linalg.norm(u, dim=1).view(-1, 1)
    a = torch.zeros_like(b)
    a[:, 0] = 1
    v = torch.linalg.cross(a, b)
    c = torch.linalg.vecdot(a, b)
    s = torch.linalg.norm(v, dim=1)
    k = torch.zeros((len(v), 3, 3), dtype=u.dtype)
    k[:, 0, 1], k[:, 0, 2] = -v[:, 2], v[:, 1]
    k[:, 1, 0], k[:, 1, 2] = v[:, 2], -v[:, 0]
    k[:, 2, 0], k[:, 2, 1] = -v[:, 1], v[:, 0]
    r = torch.eye(3, dtype=u.dtype).expand(len(v), 3, 3) + k
    r = r + k.matmul(k) * ((1 - c) / (s**2))[:, None, None]
    r[s == 0] = torch.eye(3, dtype=u.dtype)
    return r


def edge_polarizability_vectors(emb: torch.Tensor, unit: torch.Tensor) -> torch.Tensor:
    """``[E,12],[E,3] -> [E,6]``: six diag(p,q,q) tensors rotated onto the bond and
    masked to one component each (ramannoodle/pmodel/torch/_gnn.py:354-415)."""
    rot = rotations_ex_to(unit)
    inv = torch.linalg.inv(rot)
    masks = [
        [[0, 1, 0], [1, 0, 0], [0, 0, 0]], [[0, 0, 1], [0, 0, 0], [1, 0, 0]],
        [[0, 0, 0], [0, 0, 1], [0, 1, 0]], [[1, 0, 0], [0, 0, 0], [0, 0, 0]],
        [[0, 0, 0], [0, 1, 0], [0, 0, 0]], [[0, 0, 0], [0, 0, 0], [0, 0, 1]],
    ]
    total = torch.zeros((emb.size(0), 3, 3), dtype=emb.dtype)
    for m in range(6):
        p, q = emb[:, 2 * m], emb[:, 2 * m + 1]
        a = torch.zeros((emb.size(0), 3, 3), dtype=emb.dtype)
        a[:, 0, 0], a[:, 1, 1], a[:, 2, 2] = p, q, q
        total = total + (rot @ a @ inv) * torch.tensor(masks[m], dtype=emb.dtype)
    return total[:, TENSOR_TO_VEC[0], TENSOR_TO_VEC[1]]


# ----------------------------------------------------------------------------- stages
def batch_indices(model: OracleModel, s: int):
    """Replicate topology for ``s`` disconnected graphs
    (ramannoodle/pmodel/torch/_utils.py:195-223; _gnn.py:596-600)."""
    if s not in model._trip_cache:
        n, e, t = model.num_atoms, model.num_edges, model.num_triplets
        off_e = torch.arange(s).repeat_interleave(e)
        off_t = torch.arange(s).repeat_interleave(t)
        tr = [x.repeat(s) for x in model.trip]
        tr[0] = tr[0] + off_e * n
        tr[1] = tr[1] + off_e * n
        for q in (2, 3, 4):
            tr[q] = tr[q] + off_t * n
        for q in (5, 6):
            tr[q] = tr[q] + off_t * e
        model._trip_cache = {s: (tuple(tr), off_e)}
    return model._trip_cache[s]


def geometry(model: OracleModel, positions: torch.Tensor, faithful: bool, lattices=None):
    """Per-frame unit vectors ``[S*E,3]`` and distances ``[S*E,1]``
    (ramannoodle/pmodel/torch/_gnn.py:602-615, _utils.py:78-84).  ``lattices`` ``[S,3,3]``:
    one lattice per sample, as ``forward`` receives it (_gnn.py:607-610); default: the
    reference structure's lattice for every sample (what ``calc_polarizabilities`` passes,
    _gnn.py:694-700)."""
    s = positions.size(0)
    a, b = model.edges[1], model.edges[2]
    if lattices is None:
        lat = torch.from_numpy(model.lattice).type(positions.dtype).unsqueeze(0).expand(s, -1, -1)
    else:
        lat = torch.as_tensor(lattices).type(positions.dtype)
    if faithful:
        cart = min_image_cart(lat, positions)
        dist_m = torch.sqrt(torch.sum(cart**2, dim=-1))
        g = torch.arange(s).repeat_interleave(a.numel())
        aa, bb = a.repeat(s), b.repeat(s)
        vec = cart[g, aa, bb]
        dist = dist_m[g, aa, bb].view(-1, 1)
    else:
        d = positions[:, b, :] - positions[:, a, :]
        m = d % 1
        d = torch.where(m > 0.5, m - 1, m)
        vec = d.matmul(lat).reshape(-1, 3)
        dist = torch.sqrt(torch.sum(vec**2, dim=-1)).view(-1, 1)
    unit = vec / torch.linalg.norm(vec, dim=-1)[:, None]
    return unit, dist


def node_embedding(model: OracleModel, s: int, atomic_numbers=None):
    """Embedding -> ssp -> Linear -> ssp -> Linear (ramannoodle/pmodel/torch/_gnn.py:508-514,
    541-557, 642-643).  ``atomic_numbers`` ``[S,N]``: the species of every sample, flattened as
    ``_convert_to_atom_type`` does (``_gnn.py:541-557``); ``None`` = the reference structure's."""
    sd = model.sd
    if atomic_numbers is None:
        z = torch.from_numpy(model.atomic_numbers.astype(np.int64)).repeat(s)
    else:
        z = torch.as_tensor(np.asarray(atomic_numbers)).long().reshape(-1)
    types = model.atom_type_map[z].long()
    x = sd["_node_embedding.0.weight"][types]
    x = lin(sd, "_node_embedding.2", ssp(x))
    return lin(sd, "_node_embedding.4", ssp(x))


def gaussian_rbf(model: OracleModel, dist: torch.Tensor):
    """exp(coef * (d - mu)^2) (ramannoodle/pmodel/torch/_gnn.py:81-82)."""
    x = dist.view(-1, 1) - model.sd["_edge_embedding.offset"].view(1, -1)
    return torch.exp(model.coefficient * x.pow(2))


def node_block(sd, p, node, edge, i):
    """ramannoodle/pmodel/torch/_gnn.py:141-151."""
    pre = f"_node_blocks.{p}."
    c1 = torch.cat([node[i], edge], dim=1)
    c1 = lnorm(sd, pre + "c1_norm", lin(sd, pre + "c1_linear", c1))
    agg = seg_sum(gate(c1), i, node.size(0))
    return (node + lnorm(sd, pre + "final_norm", agg)).tanh()


def edge_block(sd, p, node, edge, i, j, t_i, t_j, t_k, slot5, slot6):
    """ramannoodle/pmodel/torch/_gnn.py:223-228 (c2), 270-291 (c3), 351 (residual).
    ``slot5``/``slot6`` are the 6th/7th entries of the triplet tuple, passed positionally
    into the parameters the reference names ``index_ji``/``index_kj`` (_gnn.py:650)."""
    pre = f"_edge_blocks.{p}."
    c2 = node[i] * node[j]
    c2 = lnorm(sd, pre + "c2_norm_1", lin(sd, pre + "c2_linear", c2))
    c2 = lnorm(sd, pre + "c2_norm_2", gate(c2))
    c3 = torch.cat([node[t_i], node[t_j], node[t_k], edge[slot5], edge[slot6]], dim=1)
    c3 = lnorm(sd, pre + "c3_norm_1", lin(sd, pre + "c3_linear", c3))
    c3 = seg_sum(gate(c3), slot5, edge.size(0))
    c3 = lnorm(sd, pre + "c3_norm_2", c3)
    return (edge + c2 + c3).tanh()


def readout_mlp(sd, edge, train: bool = False):
    """Linear -> BatchNorm1d -> ssp -> Linear -> ssp -> Linear(12)
    (ramannoodle/pmodel/torch/_gnn.py:532-539).  ``train=True`` normalises with the batch
    statistics over all rows, as ``model.train()`` does in ``_train.py:63``."""
    pre = "_to_polarizability_embedding."
    x = lin(sd, pre + "0", edge)
    if train:
        x = F.batch_norm(x, None, None, sd[pre + "1.weight"], sd[pre + "1.bias"], True, 0.0, 1e-5)
    else:
        x = F.batch_norm(x, sd[pre + "1.running_mean"], sd[pre + "1.running_var"],
                         sd[pre + "1.weight"], sd[pre + "1.bias"], False, 0.0, 1e-5)
    x = lin(sd, pre + "3", ssp(x))
    return lin(sd, pre + "5", ssp(x))


def forward(model: OracleModel, positions, faithful: bool = True, stages: dict | None = None,
            grad: bool = False, train: bool = False, lattices=None, atomic_numbers=None):
    """Standardised polarizability 6-vectors ``[S,6]``
    (ramannoodle/pmodel/torch/_gnn.py:617-665).  ``grad=True`` keeps the autograd graph (used
    to check the device's reverse-mode Jacobian d alpha / d r)."""
    positions = torch.as_tensor(positions).type(model.dtype)
    s = positions.size(0)
    e = model.num_edges
    sd = model.sd
    with torch.set_grad_enabled(grad):
        unit, dist = geometry(model, positions, faithful, lattices)
        node = node_embedding(model, s, atomic_numbers)
        edge = gaussian_rbf(model, dist)
        trip, off_e = batch_indices(model, s)
        if stages is not None:
            stages.update(unit=unit, dist=dist, node0=node, edge0=edge)
        for p in range(model.passes):
            node = node_block(sd, p, node, edge, trip[0])
            edge = edge_block(sd, p, node, edge, *trip)
            if stages is not None:
                stages[f"node{p + 1}"] = node
                stages[f"edge{p + 1}"] = edge
        emb = readout_mlp(sd, edge, train)
        vec = edge_polarizability_vectors(emb, unit)
        if stages is not None:
            stages["pol_emb"] = emb
            stages["edge_vec"] = vec
        if faithful:  # per-structure masked sum over ALL rows (_gnn.py:659-664)
            out = torch.zeros((s, 6), dtype=model.dtype)
            graph = off_e.view(1, -1)
            for si in range(s):
                mask = (graph == si).T
                out[si] = torch.sum(vec * mask, dim=0) / torch.sum(mask)
        else:
            out = vec.view(s, e, 6).sum(dim=1) / e
    return out


def calc_polarizabilities(model: OracleModel, positions_batch: np.ndarray,
                          faithful: bool = True, sub_batch: int = 100) -> np.ndarray:
    """``float64[S,N,3] -> float64[S,3,3]`` with the reference's 100-frame sub-batching,
    6->3x3 expansion and de-standardisation (ramannoodle/pmodel/torch/_gnn.py:683-721)."""
    if positions_batch.ndim != 3 or positions_batch.shape[1:] != (model.num_atoms, 3):
        raise ValueError("positions_batch has wrong shape")
    out = torch.zeros((positions_batch.shape[0], 3, 3), dtype=model.dtype)
    for lo in range(0, positions_batch.shape[0], sub_batch):
        chunk = positions_batch[lo:lo + sub_batch]
        vec = forward(model, torch.tensor(chunk).type(model.dtype), faithful)
        out[lo:lo + chunk.shape[0]] = vec[:, VEC_TO_TENSOR]
    return out.numpy() * model.std + model.mean


# ----------------------------------------------------------------------------- Raman
def raman_tensors_fd(model: OracleModel, ref_positions, displacements, delta=0.001):
    """(alpha(r + delta d) - alpha(r - delta d)) / delta  -- divides by delta, not 2 delta
    (ramannoodle/dynamics/_phonon.py:93-106; constants.py:248)."""
    out = []
    for d in displacements:
        plus = calc_polarizabilities(model, np.array([ref_positions + d * delta]))[0]
        minus = calc_polarizabilities(model, np.array([ref_positions - d * delta]))[0]
        out.append((plus - minus) / delta)
    return np.array(out)


def jacobian(model: OracleModel, positions_one: np.ndarray) -> np.ndarray:
    """``d vec6_k / d x`` at one structure by torch autograd, ``[6, N, 3]`` (float64 model)."""
    x = torch.tensor(positions_one[None], dtype=model.dtype, requires_grad=True)
    out = forward(model, x, faithful=False, grad=True)
    rows = []
    for k in range(6):
        (g,) = torch.autograd.grad(out[0, k], x, retain_graph=True)
        rows.append(g[0].numpy().copy())
    return np.array(rows)


def train_gradients(model: OracleModel, positions: np.ndarray, targets: np.ndarray, lattices=None, atomic_numbers=None):
    """One training step's forward/backward (``_train.py:63-73`` with ``MSELoss``): returns
    ``(out [S,6], loss, {parameter name: gradient})`` by torch autograd.  ``lattices`` ``[S,3,3]`` /
    ``atomic_numbers`` ``[S,N]``: per-sample inputs of ``forward`` (``_gnn.py:603-611, 541-557``)."""
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(
        ("offset", "running_mean", "running_var")) else v) for k, v in model.sd.items()}
    m = OracleModel(model.lattice, model.atomic_numbers, model.edges, model.trip, model.atom_type_map,
                    sd, model.coefficient, model.fn, model.fe, model.passes, model.mean, model.std,
                    model.dtype)
    out = forward(m, positions, faithful=False, grad=True, train=True, lattices=lattices, atomic_numbers=atomic_numbers)
    loss = F.mse_loss(out, torch.as_tensor(targets).type(out.dtype))
    loss.backward()
    grads = {k: v.grad.detach().numpy().copy() for k, v in sd.items()
             if v.is_floating_point() and v.requires_grad}
    return out.detach().numpy(), float(loss.detach()), grads
