"""Shared helpers for the tests (builders for product models from golden fixtures and
synthetic workloads)."""
import numpy as np

from ramannoodle_amd.pmodel import PotGNN
from ramannoodle_amd.structure import ReferenceStructure


def product_model_from_golden(g, mean=None, stddev=None, **kw):
    """Build the device PotGNN from a golden fixture's structure, hyper-parameters and
    state dict (the graph is rebuilt by the product's own host code); ``mean`` / ``stddev``
    override the fixture's de-standardisation tensors."""
    hp = g["hp"]
    ref = ReferenceStructure([int(z) for z in g["atomic_numbers"]], g["lattice"], g["positions"])
    model = PotGNN(ref, float(hp[0]), int(hp[1]), int(hp[2]), int(hp[3]), float(hp[4]),
                   float(hp[5]), g["mean"] if mean is None else mean, g["std"] if stddev is None else stddev, **kw)
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd/")}
    model.load_state_dict(sd)
    return model
