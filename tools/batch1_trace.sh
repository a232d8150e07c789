#!/bin/bash
# usage: bash tools/batch1_trace.sh perf|parity  -- rocprofv3 kernel trace of 200 one-structure calls; prints the last calls' kernels
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
out="$root/gpurun_out/batch1_$1"
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -o b1 -- python3 "$root/tools/batch1_trace.py" "$1" 200 > "$out/run.log" 2>&1
cd "$root"
grep "us per call" "$out/run.log"
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-45:]
prev = None
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-56s dur %7.1f us  gap %7.1f  grid %s" % (r["Kernel_Name"][:56], (en - st) / 1e3, (st - prev) / 1e3 if prev else 0, r["Grid_Size_X"]))
    prev = en
PY
