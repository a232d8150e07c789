#!/bin/bash
# Profile one bench.py configuration on the GPU box: rocprofv3 kernel stats, HBM traffic (two PMC
# passes) and SQ utilisation counters (separate PMC passes; never combined with runtime traces).
#   usage: bash tools/profile.sh TAG [bench.py arguments...]
# (bench.py runs with --no-cpu --no-extras: every profiled launch belongs to the timed steps, so the PMC passes
#  -- one step, no warm-up -- cover exactly `frames` structures)
# Writes gpurun_out/TAG/{stats,fetch,write,sq1,sq2,sq3}/ and the summaries
# gpurun_out/TAG/{kernel_summary.txt,pmc_traffic.txt,sq_counters.txt}; copy what should be judged
# into profiles/rNN/.
set -uo pipefail
tag="$1"; shift
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
out="$root/gpurun_out/$tag"
mkdir -p "$out"
export TMPDIR=/tmp
cd "$root"
args=("$@")
run() {  # name, rocprof options..., then "--" and bench options
  local name="$1"; shift
  rocprofv3 "$@" --output-format csv -d "$out/$name" -o "$name" -- python3 "$root/bench.py" --no-cpu --no-extras "${args[@]}" "${extra[@]}" \
    > "$out/$name.log" 2>&1 || { echo "rocprofv3 pass $name failed"; tail -5 "$out/$name.log"; return 1; }
}
extra=(--steps 3 --warmup 1)
run stats --kernel-trace --stats || exit 1
extra=(--steps 1 --warmup 0)
run fetch --pmc FETCH_SIZE --kernel-trace || exit 1
run write --pmc WRITE_SIZE --kernel-trace || exit 1
run sq1 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace || exit 1
run sq2 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace || exit 1
f() { find "$out/$1" -name "*$2" | head -1; }
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu ${args[*]} --steps 3 --warmup 1"
  python3 profiles/summarize.py "$(f stats _kernel_stats.csv)"; } > "$out/kernel_summary.txt"
{ echo "# rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --no-cpu ${args[*]} --steps 1 --warmup 0"
  python3 profiles/pmc_traffic.py "$(f fetch _counter_collection.csv)" "$(f write _counter_collection.csv)"; } > "$out/pmc_traffic.txt"
{ echo "# two SQ counter passes (profiles/sq_summary.py), bench.py --no-cpu ${args[*]} --steps 1 --warmup 0"
  python3 profiles/sq_summary.py "$(f sq1 _counter_collection.csv)" "$(f sq2 _counter_collection.csv)"; } > "$out/sq_counters.txt"
cp "$(f stats _kernel_stats.csv)" "$out/kernel_stats.csv"
grep '^{' "$out/stats.log" | tail -1 > "$out/bench_line.json"
cat "$out/kernel_summary.txt" "$out/pmc_traffic.txt" "$out/sq_counters.txt"
