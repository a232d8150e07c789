#!/bin/bash
# Everything the round's committed measurements come from, in one GPU-box call (outputs under gpurun_out/).
#   usage: bash tools/round_end_gpu.sh
set -uo pipefail
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$root"
mkdir -p gpurun_out/final_r06
python -m pytest tests -x -q -m gpu > gpurun_out/final_r06/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee gpurun_out/final_r06/pytest_rc.txt
tail -n 3 gpurun_out/final_r06/pytest_gpu.log
bash tools/profile.sh final_r06_perf > gpurun_out/final_r06/profile_perf.log 2>&1; echo "profile perf rc=$?"
bash tools/profile.sh final_r06_parity --hparams parity > gpurun_out/final_r06/profile_parity.log 2>&1; echo "profile parity rc=$?"
export TMPDIR=/tmp
( cd /tmp && RN_PROBE_STEPS=12 RN_PROBE_256=1 RN_PROBE_DEVICE=1 rocprofv3 --kernel-trace -d "$root/gpurun_out/final_r06/train_prof" -o train -- \
  python3 "$root/tools/train_probe.py" perf 32 > "$root/gpurun_out/final_r06/train_probe_device.log" 2>&1 ); echo "train profile rc=$?"
# kernel table + per-step lines, and the launches of the last step (what profiles/rNN/train_step_summary.txt is assembled from)
python3 tools/train_summary.py gpurun_out/final_r06/train_prof/train_results.db gpurun_out/final_r06/train_probe_device.log > gpurun_out/final_r06/train_summary.txt 2>&1
python3 tools/step_trace.py gpurun_out/final_r06/train_prof/train_results.db > gpurun_out/final_r06/train_step_trace.txt 2>&1
RN_PROBE_STEPS=12 RN_PROBE_256=1 python3 tools/train_probe.py perf 32 > gpurun_out/final_r06/train_probe_host.log 2>&1
RN_CONFIG5_FULL=1 python3 tools/run_configs.py > gpurun_out/final_r06/other_configs.txt 2> gpurun_out/final_r06/other_configs.err; echo "run_configs rc=$?"
python3 tools/spectrum_bench.py > gpurun_out/final_r06/spectrum_bench.txt 2>&1; echo "spectrum rc=$?"
python3 bench.py > gpurun_out/final_r06/bench_default.json 2> gpurun_out/final_r06/bench_default.err; echo "bench rc=$?"
tail -n 1 gpurun_out/final_r06/bench_default.json
