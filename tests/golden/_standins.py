"""Stand-ins for the third-party symbols the reference imports but this image lacks.

Used ONLY by ``make_golden.py`` (fixture generation, run in the build container where
``/root/reference`` is mounted).  Nothing here ships in the product or runs on the
GPU box.  The reference's own hot-path code (``ramannoodle/pmodel/torch/_gnn.py``,
``_utils.py``, ``dynamics/``, ``spectrum/``) executes unmodified on top of these.

What each stand-in replaces and how exactly its semantics are known:

* ``spglib.get_symmetry``            -- identity operation only.  PotGNN never reads
  symmetry; this merely lets ``ReferenceStructure.__init__`` complete
  (``ramannoodle/structure/_reference.py:114-125``).
* ``defusedxml.ElementTree``         -- stdlib ``xml.etree.ElementTree`` (import only).
* ``torch_geometric.nn.inits.reset`` -- calls ``reset_parameters`` recursively.
* ``...schnet.ShiftedSoftplus``      -- ``softplus(x) - log 2`` (unambiguous).
* ``torch_geometric.utils.scatter``  -- ``zeros.index_add_`` for ``reduce="sum"``
  (unambiguous up to fp32 summation order).
* ``...dimenet.triplets``            -- restated from the published PyG source
  (2.3-2.6): ordering and the 7-tuple return order
  ``(col, row, idx_i, idx_j, idx_k, idx_kj, idx_ji)``.  NOT verifiable here (no PyG
  install, no network) => "parity unpinned" at the triplet *ordering*; the triplet
  *set* is fixed by the graph and ordering only changes fp32 summation order.
"""
from __future__ import annotations

import math
import sys
import types
import xml.etree.ElementTree as _ET

import numpy as np
import torch


def _get_symmetry(cell, symprec=1e-5, angle_tolerance=-1.0):
    num_atoms = len(cell[1])
    return {
        "rotations": np.eye(3, dtype=np.int32)[None],
        "translations": np.zeros((1, 3)),
        "equivalent_atoms": np.arange(num_atoms, dtype=np.int32),
    }


def _reset(module):
    if hasattr(module, "reset_parameters"):
        module.reset_parameters()
    else:
        for child in module.children() if hasattr(module, "children") else []:
            _reset(child)


class _ShiftedSoftplus(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.shift = math.log(2.0)

    def forward(self, x):
        return torch.nn.functional.softplus(x) - self.shift


def _scatter(src, index, dim=0, dim_size=None, reduce="sum"):
    assert dim == 0 and reduce == "sum"
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.index_add_(0, index, src)


def _triplets(edge_index, num_nodes):
    row, col = edge_index  # j -> i
    num_edges = row.numel()
    # SparseTensor(row=col, col=row, value=arange(E)): entries sorted by (col, row).
    order = torch.argsort(col * num_nodes + row, stable=True)
    sp_row = col[order]
    sp_col = row[order]
    sp_val = order
    rowptr = torch.zeros(num_nodes + 1, dtype=torch.long)
    rowptr[1:] = torch.cumsum(torch.bincount(sp_row, minlength=num_nodes), 0)
    # adj_t[row]: for every edge e (in order) the sparse row of node row[e].
    counts = rowptr[row + 1] - rowptr[row]
    idx_ji_full = torch.repeat_interleave(torch.arange(num_edges), counts)
    starts = rowptr[row]
    offs = torch.arange(int(counts.sum())) - torch.repeat_interleave(
        torch.cumsum(counts, 0) - counts, counts
    )
    pos = torch.repeat_interleave(starts, counts) + offs
    idx_k = sp_col[pos]
    idx_kj = sp_val[pos]
    idx_i = col[idx_ji_full]
    idx_j = row[idx_ji_full]
    mask = idx_i != idx_k
    return (
        col,
        row,
        idx_i[mask],
        idx_j[mask],
        idx_k[mask],
        idx_kj[mask],
        idx_ji_full[mask],
    )


def install():
    """Register the stand-in modules in ``sys.modules``."""

    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("spglib", get_symmetry=_get_symmetry)
    d = mod("defusedxml")
    d.ElementTree = _ET
    sys.modules["defusedxml.ElementTree"] = _ET
    mod("torch_geometric")
    mod("torch_geometric.nn")
    mod("torch_geometric.nn.inits", reset=_reset)
    mod("torch_geometric.nn.models")
    mod("torch_geometric.nn.models.schnet", ShiftedSoftplus=_ShiftedSoftplus)
    mod("torch_geometric.nn.models.dimenet", triplets=_triplets)
    mod("torch_geometric.utils", scatter=_scatter)
