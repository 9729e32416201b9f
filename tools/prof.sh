#!/bin/bash
# Per-kernel times of a command on the GPU box:  tools/prof.sh TAG python3 script.py args...
# Writes gpurun_out/TAG_kernel_stats.csv and prints calls / average us / kernel name.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- "$@" > "$R/gpurun_out/${TAG}_prof.log" 2>&1
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$f" "$R/gpurun_out/${TAG}_kernel_stats.csv"
python3 "$R/tools/kstats.py" "$f" | head -${PROF_LINES:-12}
