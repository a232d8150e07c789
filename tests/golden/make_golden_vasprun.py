"""Golden fixtures for the vasprun.xml reader: the REFERENCE's parser
(``ramannoodle/io/vasp/vasprun.py``: ``read_trajectory``, ``read_positions``, ``read_ref_structure``)
run on its own test files and on generated variants / malformed documents.  Build container only::

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_vasprun.py

Writes ``tests/golden/vasprun/*.xml`` (input data: the reference's ``test/data/TiO2/md_run_vasprun.xml``
cut to its first 6 MD frames, ``test/data/STO/vasprun.xml`` as is, generated variants) and
``tests/golden/vasprun/expected.npz`` (what the reference returned: arrays, or the exception type
and message, per function).  Fixtures are data; no reference code is copied.
"""
from __future__ import annotations

import os
import re
import shutil
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402

import _standins  # noqa: E402

_standins.install()

from ramannoodle.io.vasp import vasprun as ref  # noqa: E402

OUT = os.path.join(HERE, "vasprun")
os.makedirs(OUT, exist_ok=True)


def small(frames, atoms=("Ti", "O", "O"), potim="  0.50000000", rows=None, extra="", initial=True,
          unnamed_attr="", row_tag="v"):
    """A minimal vasprun-like document (same element layout as VASP writes)."""
    n = len(atoms)
    rng = np.random.default_rng(7)
    doc = ['<?xml version="1.0" encoding="ISO-8859-1"?>', "<modeling>", ' <generator><i name="program">vasp</i></generator>',
           " <incar>", f'  <i name="POTIM">{potim}</i>', " </incar>",
           ' <parameters><separator name="electronic"><i name="POTIM">9.0</i></separator>'
           f'<separator name="ionic" ><i type="int" name="NSW">5</i><i name="POTIM">{potim}</i></separator></parameters>',
           " <atominfo><atoms>%d</atoms><array name=\"atoms\" ><dimension dim=\"1\">ion</dimension>"
           "<field type=\"string\">element</field><field type=\"int\">atomtype</field><set>" % n
           + "".join(f"<rc><c>{a:2s}</c><c>{i + 1:4d}</c></rc>" for i, a in enumerate(atoms))
           + "</set></array></atominfo>"]

    def structure(name, k):
        pos = rows[k] if rows is not None else [" ".join(f"{v:16.8f}" for v in rng.uniform(size=3)) for _ in range(n)]
        body = "".join(f"   <{row_tag}> {r} </{row_tag}>\n" for r in pos)
        return (f' <structure{name}>\n  <crystal>\n   <varray name="basis" >\n    <v> 4.0 0.0 0.0 </v>\n'
                "    <v> 0.1 5.0 0.0 </v>\n    <v> 0.0 0.2 6.0 </v>\n   </varray>\n  </crystal>\n"
                f'  <varray name="positions" >\n{body}  </varray>\n </structure>')

    if initial:
        doc.append(structure(' name="initialpos" ', 0))
    for k in range(frames):
        doc.append(structure(unnamed_attr, k))
    doc.append(extra)
    doc.append(' <structure name="finalpos" ><varray name="positions" ><v> 0 0 0 </v></varray></structure>')
    doc.append("</modeling>")
    return "\n".join(doc) + "\n"


def cut_md(text, frames):
    """Keep the first `frames` un-named <structure> elements (and what lies between them)."""
    starts = [m.start() for m in re.finditer(r"<structure>", text)]
    if len(starts) <= frames:
        return text
    return text[: starts[frames]] + "</modeling>\n"


good = small(4)
ROWS3 = [["0.1 0.2 0.3", "0.25 +0.5 1e-1", ".75 5. -0.125"]] * 2
CASES = {
    "tio2_md": ("file", "/root/reference/test/data/TiO2/md_run_vasprun.xml", 6),
    "sto": ("file", "/root/reference/test/data/STO/vasprun.xml", None),
    # the two inputs of the reference's own exception test (test/tests/test_vasprun.py:154-200):
    # an empty <modeling> root, and a POSCAR handed over as a vasprun.xml
    "ref_malformed": ("file", "/root/reference/test/data/malformed/vasprun.xml", None),
    "ref_poscar_as_xml": ("file", "/root/reference/test/data/TiO2/POSCAR", None),
    "small": good,
    "number_forms": small(2, rows=[["1_0.5 inf -Infinity", "nan 1E2 2e-3", "1 2 3"], ["1 2 3", "4 5 6", "7 8 9"]]),
    "comments_cdata_entities": small(2, rows=[["0.1 <!-- c --> 0.2 0.3", "<![CDATA[0.4 0.5]]> 0.6", "&#48;.7 0.8 0.9"],
                                              ["1 2 3", "4 5 6", "7 8 9"]]),
    "one_frame": small(1),
    "no_frames": small(0),
    "no_initial": small(2, initial=False),
    "named_frames_only": small(2, unnamed_attr=' name="step" '),
    "self_closing_row": small(2, rows=ROWS3).replace("<v> 0.25 +0.5 1e-1 </v>", "<v/>", 1),
    "empty_row_text": small(2, rows=ROWS3).replace("<v> 0.25 +0.5 1e-1 </v>", "<v></v>", 1),
    "bad_token": small(2, rows=[["0.1 0.2 0.3", "0.1 zero 0.3", "1 2 3"], ["1 2 3", "4 5 6", "7 8 9"]]),
    "potim_missing": good.replace('<i name="POTIM">  0.50000000</i></separator></parameters>', "</separator></parameters>"),
    "potim_empty": good.replace('<i name="POTIM">  0.50000000</i></separator></parameters>',
                                '<i name="POTIM"></i></separator></parameters>'),
    "potim_bad": small(2, potim="fast"),
    "structure_without_varray": small(2, extra=" <structure><crystal></crystal></structure>"),
    "no_atominfo": good.replace("<atominfo>", "<atominfox>").replace("</atominfo>", "</atominfox>"),
    "no_lattice": good.replace('<varray name="basis" >', '<varray name="bases" >', 1),
    # ill-formed XML
    "truncated": good[: len(good) // 2],
    "mismatched_tag": good.replace("</incar>", "</incarx>"),
    "junk_after_root": good + "trailing\n",
    "not_xml": "hello\n",
    "empty": "",
    "unclosed_comment": good.replace("<incar>", "<!-- <incar>", 1),
}


def attempt(fn, path):
    try:
        return ("ok", fn(path))
    except Exception as exc:  # pylint: disable=broad-except
        return ("err", type(exc).__name__, str(exc))


expected = {}
for name, spec in CASES.items():
    path = os.path.join(OUT, f"{name}.xml")
    if isinstance(spec, tuple):
        text = open(spec[1], encoding="utf-8").read()
        if spec[2] is not None:
            text = cut_md(text, spec[2])
    else:
        text = spec
    with open(path, "w", encoding="utf-8", newline="") as f:
        f.write(text)
    os.chmod(path, 0o644)

    res = attempt(ref.read_trajectory, path)
    if res[0] == "ok":
        expected[f"{name}/traj/positions"] = res[1].positions_ts
        expected[f"{name}/traj/timestep"] = np.float64(res[1].timestep)
    else:
        expected[f"{name}/traj/error_type"], expected[f"{name}/traj/error_message"] = np.array(res[1]), np.array(res[2])
    res_p = attempt(ref.read_positions, path)
    if res_p[0] == "ok":
        expected[f"{name}/pos/positions"] = np.asarray(res_p[1], dtype=np.float64)
    else:
        expected[f"{name}/pos/error_type"], expected[f"{name}/pos/error_message"] = np.array(res_p[1]), np.array(res_p[2])
    res_s = attempt(ref.read_ref_structure, path)
    if res_s[0] == "ok":
        expected[f"{name}/ref/lattice"] = res_s[1].lattice
        expected[f"{name}/ref/positions"] = res_s[1].positions
        expected[f"{name}/ref/atomic_numbers"] = np.array(res_s[1].atomic_numbers)
    else:
        expected[f"{name}/ref/error_type"], expected[f"{name}/ref/error_message"] = np.array(res_s[1]), np.array(res_s[2])
    print(name, res[:2] if res[0] == "err" else res[1].positions_ts.shape, "|",
          res_p[:3] if res_p[0] == "err" else np.asarray(res_p[1]).shape, "|", res_s[:3] if res_s[0] == "err" else "ref ok")
np.savez_compressed(os.path.join(OUT, "expected.npz"), **expected)
