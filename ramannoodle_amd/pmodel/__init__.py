"""Polarizability models (device PotGNN only; see SURVEY.md section 2 for scope)."""
from ramannoodle_amd.pmodel.potgnn import (  # noqa: F401
    PotGNN,
    polarizability_tensors_to_vectors,
    polarizability_vectors_to_tensors,
)

__all__ = ["PotGNN", "polarizability_vectors_to_tensors", "polarizability_tensors_to_vectors"]
