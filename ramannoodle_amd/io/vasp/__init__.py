"""VASP file formats (mirrors ``ramannoodle.io.vasp``: only the trajectory side of the hot path)."""
from ramannoodle_amd.io.vasp import vasprun, xdatcar  # noqa: F401
