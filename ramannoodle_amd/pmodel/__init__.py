"""Polarizability models (device PotGNN only; see SURVEY.md section 2 for scope)."""
from ramannoodle_amd.pmodel.potgnn import (  # noqa: F401
    DeviceAdam,
    PotGNN,
    polarizability_tensors_to_vectors,
    polarizability_vectors_to_tensors,
)

from ramannoodle_amd.pmodel.train import train_single_epoch  # noqa: F401

__all__ = ["PotGNN", "DeviceAdam", "train_single_epoch", "polarizability_vectors_to_tensors",
           "polarizability_tensors_to_vectors"]
