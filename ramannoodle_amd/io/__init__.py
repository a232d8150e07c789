"""Trajectory ingest: native readers that feed the evaluator."""
from ramannoodle_amd.io import vasp  # noqa: F401
