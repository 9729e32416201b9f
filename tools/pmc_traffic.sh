#!/bin/bash
# HBM traffic of a command's kernels from the PMC counters, on the GPU box:
#   tools/pmc_traffic.sh TAG KERNEL_SUBSTRING python3 script.py args...
# Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only), summarised per kernel
# into gpurun_out/TAG_pmc_{FETCH,WRITE}_SIZE.txt (launches, median / min / max in KiB as rocprofv3 reports them;
# FETCH_SIZE counts 128-byte fabric requests as 64 bytes on gfx950: double it).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; FILTER=$2; shift 2
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL ${PMC_TIMEOUT:-240} rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$TAG/$c -o c -- "$@" > "$R/gpurun_out/${TAG}_pmc_$c.log" 2>&1
    python3 "$R/tools/pmc_summary.py" "$(find /tmp/pmc_$TAG/$c -name '*counter_collection.csv' | head -1)" "$FILTER" > "$R/gpurun_out/${TAG}_pmc_$c.txt"
    cat "$R/gpurun_out/${TAG}_pmc_$c.txt"
done
