#!/usr/bin/env python3
"""Turn the summaries tools/profile.sh wrote (gpurun_out/TAG/) into the small JSON records bench.py
reads for `roofline.traffic` / `roofline.issue_frac` (profiles/rNN/*.json).

    python tools/make_profile_json.py gpurun_out/TAG profiles/r02 PREFIX N E FN FE FRAMES PASSES

FRAMES = frames of the profiled bench step (--frames, or the config's default); the PMC passes run
`bench.py --no-cpu --no-extras --steps 1 --warmup 0` (tools/profile.sh), i.e. exactly FRAMES structures and
nothing else.  Of the instantiations of a kernel family the one that moved the most bytes is taken.  Units as
profiles/pmc_traffic.py prints them (reads doubled as MI355X_MICROARCH.md prescribes for wide coalesced
streams on gfx950)."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# which source file a kernel family lives in: bench.py replays a record only while that file is unchanged
SOURCE = {"narrow": "kernels_narrow.hip", "ps": "kernels_edge_ps.hip", "atom": "kernels_node_atom.hip",
          "fused": "kernels_fused.hip", "agg": "kernels_agg.hip"}


def stamp(family):
    rel = os.path.join("ramannoodle_amd", "csrc", SOURCE[family])
    data = open(os.path.join(ROOT, rel), "rb").read()
    return {"file": rel, "git_blob": hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()}


src, dst, prefix = sys.argv[1], sys.argv[2], sys.argv[3]
n, e, fn, fe, frames, passes = (int(v) for v in sys.argv[4:10])
structures = frames
kernels = {"edge": ("edge_block_ps_kernel", "edge_block_fused_kernel", "edge_block2_kernel", "edge_narrow_kernel", "edge_agg_kernel"),
           "node": ("node_block_atom_kernel", "node_block_fused_kernel", "node_tiled_kernel", "node_narrow_kernel", "node_agg_kernel")}
traffic = open(os.path.join(src, "pmc_traffic.txt")).read().splitlines()
sq = open(os.path.join(src, "sq_counters.txt")).read().splitlines()
os.makedirs(dst, exist_ok=True)
for kind, names in kernels.items():
    cand = []
    for line in traffic:
        m = re.match(r"(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+\d+\s+\d+$", line)
        if m and any(k in m.group(1) for k in names):
            cand.append((int(m.group(2)) * (float(m.group(3)) + float(m.group(4))), m))
    for _, m in sorted(cand, key=lambda c: -c[0])[:1]:
        kernel, launches, rd, wr = m.group(1).strip(), int(m.group(2)), float(m.group(3)), float(m.group(4))
        per = (rd + wr) * 1e6 * launches / (structures * passes)
        rec = {"kernel": kernel, "workload_shape": [n, e, fn, fe], "hbm_bytes_per_structure_pass": per,
               "read_bytes_per_structure_pass": rd * 1e6 * launches / (structures * passes),
               "write_bytes_per_structure_pass": wr * 1e6 * launches / (structures * passes),
               "source": f"{os.path.basename(src)}/pmc_traffic.txt: {launches} launches over {structures} structures x {passes} passes"}
        family = ("narrow" if ("narrow" in kernel or "tiled" in kernel) else "ps" if "block_ps" in kernel
                  else "atom" if "block_atom" in kernel else "fused" if ("fused" in kernel or "block2" in kernel) else "agg")
        rec["kernel_source"] = stamp(family)
        json.dump(rec, open(os.path.join(dst, f"{prefix}{kind}_{family}_traffic.json"), "w"), indent=1)
        print(kind, "traffic", round(per), "B per structure and pass")
        break
    for line in sq:
        f = line.split()
        if len(f) >= 10 and any(k in line for k in names):
            valu, mfma, waves = float(f[-6]), float(f[-5]), float(f[-1])
            rec = {"kernel": " ".join(f[:-9]), "workload_shape": [n, e, fn, fe], "valu_busy": valu, "mfma_busy": mfma,
                   "valu_plus_mfma_busy": valu + mfma, "waves_per_simd": waves, "wait_any": float(f[-4]),
                   "source": f"{os.path.basename(src)}/sq_counters.txt"}
            family = ("narrow" if ("narrow" in line or "tiled" in line) else "ps" if "block_ps" in line
                      else "atom" if "block_atom" in line else "fused" if ("fused" in line or "block2" in line) else "agg")
            rec["kernel_source"] = stamp(family)
            json.dump(rec, open(os.path.join(dst, f"{prefix}{kind}_{family}_issue.json"), "w"), indent=1)
            print(kind, "issue", valu + mfma)
            break
