#!/bin/bash
# Per-kernel average durations of one tools/plan_ab.py invocation (rocprofv3 kernel trace).
#   tools/kernel_times.sh <tag> <plan_ab arguments...>      -> gpurun_out/kt_<tag>.txt
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/kt_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python3 "$root/tools/plan_ab.py" "$@" > "$out/run.log" 2>&1
f=$(find "$out" -name '*kernel_stats.csv' | head -1)
python3 - "$f" > "$root/gpurun_out/kt_$tag.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{float(r['AverageNs'])/1e3:10.1f} us avg  x{r['Calls']:>6}  {r['Name'][:150]}")
PY
grep -v amdgpu.ids "$out/run.log" >> "$root/gpurun_out/kt_$tag.txt" || true
rm -rf "$out"
