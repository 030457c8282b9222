#!/bin/bash
# Regenerates profiles/<tag>_* on a GPU box: kernel-trace stats of the default (overlapped) step and of the step with the
# side stream off (one kernel on the chip at a time), and the per-kernel HBM traffic from separate FETCH_SIZE / WRITE_SIZE passes.
#   bash tools/profile_step.sh r02
set -e
TAG="${1:-rXX}"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$R/gpurun_out/prof" "$R/profiles"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1 /tmp/kt2 /tmp/pf /tmp/pw
rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt --output-format csv -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > /tmp/kt1.log 2>&1
cp "$(find /tmp/kt1 -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/prof/${TAG}_step_kernel_stats.csv"
echo "default run done"
GIPVIT_DW_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt --output-format csv -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > /tmp/kt2.log 2>&1
cp "$(find /tmp/kt2 -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/prof/${TAG}_step_kernel_stats_exclusive.csv"
echo "exclusive run done"
GIPVIT_DW_STREAM=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o p --output-format csv -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pf.log 2>&1
echo "fetch pass done"
GIPVIT_DW_STREAM=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o p --output-format csv -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pw.log 2>&1
echo "write pass done"
python3 "$R/tools/pmc_traffic.py" /tmp/pf /tmp/pw "$R/gpurun_out/prof/${TAG}_hbm_traffic_per_kernel.json"
