"""Argument checks and error types shared by the public API.

Mirrors the error behaviour of ``ramannoodle/exceptions.py:57-105`` (same exception
classes and message formats) because callers and the reference's tests rely on them:
``Trajectory``/``Phonons.get_raman_spectrum`` re-raise ``ValueError`` from a model as
"... are incompatible" (``dynamics/_trajectory.py:81-88``, ``dynamics/_phonon.py:94-105``).
"""
from __future__ import annotations

from typing import Any, Sequence


class UserError(Exception):
    """The user has done something they shouldn't (``ramannoodle/exceptions.py:33``)."""


class InvalidFileException(Exception):
    """File cannot be read, likely due to an invalid or unexpected format
    (``ramannoodle/exceptions.py:12``)."""


class DeviceError(RuntimeError):
    """The HIP library is missing or a device call failed."""


def shape_string(shape: Sequence[int | None]) -> str:
    """``(3,_,3)`` style rendering; ``None`` -> ``_``; 1-tuples keep the trailing comma."""
    parts = ["_" if dim is None else str(dim) for dim in shape]
    if len(parts) == 1:
        return "(" + parts[0] + ",)"
    return "(" + ",".join(parts) + ")"


def get_type_error(name: str, value: Any, correct_type: str) -> TypeError:
    return TypeError(f"{name} should have type {correct_type}, not {type(value).__name__}")


def get_shape_error(name: str, array: Any, desired_shape: str) -> ValueError:
    return ValueError(f"{name} has wrong shape: {shape_string(array.shape)} != {desired_shape}")


def verify_ndarray_shape(name: str, array: Any, shape: Sequence[int | None]) -> None:
    """Raise ``ValueError`` on a shape mismatch, ``TypeError`` for a non-array."""
    try:
        ok = len(shape) == array.ndim and all(
            want is None or have == want for have, want in zip(array.shape, shape)
        )
    except AttributeError as exc:
        raise get_type_error(name, array, "ndarray") from exc
    if not ok:
        raise get_shape_error(name, array, shape_string(shape))
