"""Host-side construction of the frozen reference graph (one-time, integer results).

``radius_graph_pbc`` restates ``ramannoodle/pmodel/torch/_utils.py:118-141`` in numpy
with the same float32 operation order, so edge lists agree bit for bit away from
exact cutoff ties.  ``reference_order_triplets`` lists the edge triplets in the order
``torch_geometric.nn.models.dimenet.triplets`` produces (``_utils.py:153-168``); the
device kernels never read such lists (they enumerate triplets implicitly from the
sorted edge list) -- this exists for index-parity checks and for exporting a model
back to the reference.
"""
from __future__ import annotations

import numpy as np


def radius_graph_pbc(lattice: np.ndarray, positions: np.ndarray, cutoff: float,
                     dtype=np.float32) -> np.ndarray:
    """Edges ``int64[2,E]`` = (a, b), sorted by (a, b): pairs with minimum-image
    distance ``<= cutoff`` and ``a != b``."""
    lat = np.asarray(lattice).astype(dtype)
    pos = np.asarray(positions).astype(dtype)
    disp = pos[None, :, :] - pos[:, None, :]  # [a, b] = x_b - x_a
    rem = np.remainder(disp, dtype(1))
    disp = np.where(rem > dtype(0.5), rem - dtype(1), rem)
    cart = np.matmul(disp, lat)
    dist = np.sqrt(np.sum(cart**2, axis=-1, dtype=dtype))
    adjacency = dist <= dtype(cutoff) if dtype == np.float32 else dist <= cutoff
    np.fill_diagonal(adjacency, False)
    a, b = np.nonzero(adjacency)
    return np.stack([a, b]).astype(np.int64)


def radius_graph_pbc_device(lattice: np.ndarray, positions: np.ndarray, cutoff: float,
                            device: int = 0) -> np.ndarray:
    """Same edge list from the HIP kernel (``rn_potgnn_radius_graph``): the N^2 pair test
    runs on the device, the flag compaction on the host."""
    import ctypes as C

    from ramannoodle_amd import _lib

    lib = _lib.load()
    lat = np.ascontiguousarray(lattice, dtype=np.float64)
    pos = np.ascontiguousarray(positions, dtype=np.float64)
    n = pos.shape[0]
    adjacency = np.zeros((n, n), dtype=np.uint8)
    rc = lib.rn_potgnn_radius_graph(C.c_void_p(lat.ctypes.data), C.c_void_p(pos.ctypes.data), n,
                                    float(cutoff), int(device), C.c_void_p(adjacency.ctypes.data))
    _lib.check(rc, None, "rn_potgnn_radius_graph")
    a, b = np.nonzero(adjacency)
    return np.stack([a, b]).astype(np.int64)


def atom_type_map(atomic_numbers) -> np.ndarray:
    """``int32[119]`` atomic number -> type index, -1 when absent.  Type indices follow
    Python ``set`` iteration order, as the reference does
    (``ramannoodle/pmodel/torch/_gnn.py:502-506``)."""
    table = np.full(119, -1, dtype=np.int32)
    for atom_type, atomic_number in enumerate(set(int(z) for z in atomic_numbers)):
        table[atomic_number] = atom_type
    return table


def reference_order_triplets(edges: np.ndarray, num_nodes: int):
    """The 7-tuple ``(col, row, idx_i, idx_j, idx_k, idx_kj, idx_ji)`` for
    ``row, col = edges`` (edge j->i), triplets k->j->i sorted by (j->i edge, k)."""
    row, col = edges[0], edges[1]
    num_edges = row.size
    order = np.argsort(col * num_nodes + row, kind="stable")  # edges grouped by destination
    in_ptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.cumsum(np.bincount(col, minlength=num_nodes), out=in_ptr[1:])
    counts = in_ptr[row + 1] - in_ptr[row]  # edges entering j = row[e]
    idx_ji = np.repeat(np.arange(num_edges), counts)
    first = np.repeat(np.cumsum(counts) - counts, counts)
    pos = np.repeat(in_ptr[row], counts) + (np.arange(idx_ji.size) - first)
    idx_kj = order[pos]
    idx_k = row[idx_kj]
    idx_i, idx_j = col[idx_ji], row[idx_ji]
    keep = idx_i != idx_k
    return col, row, idx_i[keep], idx_j[keep], idx_k[keep], idx_kj[keep], idx_ji[keep]
