#!/bin/bash
# LDS behaviour of the headline kernel (BASELINE config 2) from the SQ counters, on the GPU box:
#   tools/pmc_lds.sh TAG   ->  gpurun_out/TAG_cfg2_pmc_lds.txt
# One rocprofv3 --pmc pass (six SQ counters: within the eight the block holds) over a short run of bench.py.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-r03}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/plds_$TAG
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
    --output-format csv -d /tmp/plds_$TAG -o c -- python3 "$R/bench.py" --steps 20 --warmup 5 --settle 0 --no-cpu --no-secondary > /dev/null 2>&1
python3 "$R/tools/pmc_summary.py" "$(find /tmp/plds_$TAG -name '*counter_collection.csv' | head -1)" fft_lds > "$R/gpurun_out/${TAG}_cfg2_pmc_lds.txt"
cat "$R/gpurun_out/${TAG}_cfg2_pmc_lds.txt"
