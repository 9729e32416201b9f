#!/bin/bash
# rocprofv3 evidence for the headline workload (BASELINE config 2), on the GPU box:
#   tools/profile_cfg2.sh TAG    ->  gpurun_out/TAG_cfg2_kernel_stats.csv, TAG_cfg2_pmc_{fetch,write}.txt
# Kernel times come from --kernel-trace --stats over the default bench run; HBM traffic from two
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over a short run of the same command.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG/stats -o s -- python3 "$R/bench.py" --no-cpu --no-secondary > "$R/gpurun_out/${TAG}_cfg2_bench.json" 2> /dev/null
cp "$(find /tmp/p_$TAG/stats -name '*kernel_stats.csv' | head -1)" "$R/gpurun_out/${TAG}_cfg2_kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d /tmp/p_$TAG/$c -o c -- python3 "$R/bench.py" --steps 20 --warmup 5 --settle 0 --no-cpu --no-secondary > /dev/null 2>&1
    python3 "$R/tools/pmc_summary.py" "$(find /tmp/p_$TAG/$c -name '*counter_collection.csv' | head -1)" fft_lds > "$R/gpurun_out/${TAG}_cfg2_pmc_$c.txt"
done
head -3 "$R/gpurun_out/${TAG}_cfg2_kernel_stats.csv"
cat "$R/gpurun_out/${TAG}_cfg2_pmc_FETCH_SIZE.txt" "$R/gpurun_out/${TAG}_cfg2_pmc_WRITE_SIZE.txt"
