"""Golden fixtures for the trajectory reader: the REFERENCE's XDATCAR parser
(``ramannoodle/io/vasp/xdatcar.py::read_positions_ts``) run on its own test file and on
hand-made variants / malformed inputs.  Build container only (``/root/reference`` mounted)::

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_io.py

Writes ``tests/golden/xdatcar/*.XDATCAR`` (input data: the reference's ``test/data/STO/XDATCAR``
and generated variants) and ``tests/golden/xdatcar/expected.npz`` (what the reference returned:
positions, or the exception type and message).  Fixtures are data; no reference code is copied.
"""
from __future__ import annotations

import os
import shutil
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

import _standins  # noqa: E402

_standins.install()

from ramannoodle.io.vasp.xdatcar import read_positions_ts  # noqa: E402

OUT = os.path.join(HERE, "xdatcar")
os.makedirs(OUT, exist_ok=True)

HEADER = "generated\n   1.5\n  4.0 0.0 0.0\n  0.1 5.0 0.0\n  0.0 0.2 6.0\n  Ti O\n  1 2\n"
ROWS = ["  0.1 0.2 0.3\n", " 0.25  +0.5 1e-1  T T F\n", "\t.75 5.  -0.125\n"]


def frame(label, rows=ROWS):
    return label + "".join(rows)


CASES = {
    "sto": None,  # the reference's own test file
    "variants": HEADER + frame("Direct configuration=     1\n") + frame("direct\n") +
                frame("Cartesian configuration= 3\n", ["  0.4 0.5 0.6\n", " 1.0 2.0 3.0\n", "0 0 0\n"]) +
                frame("Selective dynamics\nDirect\n") + frame("D\r\n", [r.replace("\n", "\r\n") for r in ROWS]),
    "ends_on_blank_label": HEADER + frame("Direct\n") + "\n" + frame("Direct\n"),
    "ends_on_indented_label": HEADER + frame("Direct\n") + frame(" Direct\n"),
    "no_trailing_newline": HEADER + frame("Direct\n") + "Direct\n" + "".join(ROWS)[:-1],
    "underscores_inf": HEADER + frame("Direct\n", ["1_0.5 inf -Infinity\n", "nan 1E2 2e-3\n", "1 2 3\n"]),
    "no_frames": HEADER,
    "negative_count": HEADER.replace("  1 2\n", "  -1 3\n") + frame("Direct\n"),
    # malformed
    "bad_scale": HEADER.replace("   1.5\n", "   1.5 2\n") + frame("Direct\n"),
    "bad_lattice_token": HEADER.replace("0.1 5.0 0.0", "0.1 five 0.0") + frame("Direct\n"),
    "short_lattice": HEADER.replace("  0.0 0.2 6.0\n", "  0.0 0.2\n") + frame("Direct\n"),
    "no_symbols": HEADER.replace("  Ti O\n", "\n") + frame("Direct\n"),
    "bad_symbol": HEADER.replace("Ti O", "Ti Xx") + frame("Direct\n"),
    "count_mismatch": HEADER.replace("  1 2\n", "  1 2 3\n") + frame("Direct\n"),
    "bad_count": HEADER.replace("  1 2\n", "  1 two\n") + frame("Direct\n"),
    "bad_label": HEADER + frame("Direct\n") + frame("Fractional\n"),
    "bad_row_token": HEADER + frame("Direct\n", ["0.1 0.2 0.3\n", "0.1 0.2 zero\n", "1 2 3\n"]),
    "short_row": HEADER + frame("Direct\n", ["0.1 0.2 0.3\n", "0.1 0.2\n", "1 2 3\n"]),
    "truncated_frame": HEADER + frame("Direct\n") + frame("Direct\n", ROWS[:2]),
    "underscore_misuse": HEADER + frame("Direct\n", ["_1 0.2 0.3\n", "0.1 0.2 0.3\n", "1 2 3\n"]),
}

expected = {}
for name, text in CASES.items():
    path = os.path.join(OUT, f"{name}.XDATCAR")
    if text is None:
        shutil.copyfile("/root/reference/test/data/STO/XDATCAR", path)
    else:
        with open(path, "w", encoding="utf-8", newline="") as f:
            f.write(text)
    os.chmod(path, 0o644)
    try:
        result = read_positions_ts(path)
        expected[f"{name}/positions"] = np.asarray(result, dtype=np.float64)
        print(f"{name}: positions {np.asarray(result).shape}")
    except Exception as exc:  # pylint: disable=broad-except
        expected[f"{name}/error_type"] = np.array(type(exc).__name__)
        expected[f"{name}/error_message"] = np.array(str(exc))
        print(f"{name}: {type(exc).__name__}: {exc!r}")
np.savez(os.path.join(OUT, "expected.npz"), **expected)
