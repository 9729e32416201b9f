#!/bin/bash
# rocprofv3 evidence for the LDS-DMA kernels, on the GPU box:  tools/profile_dma.sh TAG
#   -> gpurun_out/TAG_dma_kernel_stats.txt (average launch times) and TAG_dma_pmc.txt (FETCH_SIZE / WRITE_SIZE per launch)
# 2^26 points per launch: 16384 x 4096 and 8192 x 8192 complex64, plain (flags 0) and Hann + power (flags 3).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r02}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
: > "$R/gpurun_out/${TAG}_dma_kernel_stats.txt"; : > "$R/gpurun_out/${TAG}_dma_pmc.txt"
for cfg in "16384 4096 0 0" "16384 4096 0 3" "8192 8192 0 0" "8192 8192 0 3"; do
    set -- $cfg
    rm -rf /tmp/pd_$TAG
    timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd_$TAG/stats -o s -- python3 "$R/tools/one_plan.py" $1 $2 $3 $4 default 600 > /dev/null 2>&1
    echo "# n=$1 batch=$2 flags=$4 (600 launches, clock transient of the first ~50 included)" >> "$R/gpurun_out/${TAG}_dma_kernel_stats.txt"
    python3 "$R/tools/kstats.py" "$(find /tmp/pd_$TAG/stats -name '*kernel_stats.csv' | head -1)" | grep fft_lds >> "$R/gpurun_out/${TAG}_dma_kernel_stats.txt"
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pd_$TAG/$c -o c -- python3 "$R/tools/one_plan.py" $1 $2 $3 $4 > /dev/null 2>&1
        echo "# n=$1 batch=$2 flags=$4 $c (KiB per launch)" >> "$R/gpurun_out/${TAG}_dma_pmc.txt"
        python3 "$R/tools/pmc_summary.py" "$(find /tmp/pd_$TAG/$c -name '*counter_collection.csv' | head -1)" fft_lds >> "$R/gpurun_out/${TAG}_dma_pmc.txt"
    done
done
cat "$R/gpurun_out/${TAG}_dma_kernel_stats.txt" "$R/gpurun_out/${TAG}_dma_pmc.txt"
