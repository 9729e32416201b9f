#!/bin/bash
# Kernel timeline of one fft_from_timeseries call at final2's defaults (rocprofv3 --kernel-trace), on the GPU box.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pa
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pa -o t -- python3 $R/tools/app_bench.py --loop 60 > /tmp/pa.log 2>&1 || { tail -n 5 /tmp/pa.log; exit 1; }
f=$(find /tmp/pa -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# take the last 400 kernels, split into calls by the premul kernel name
names=[r["Kernel_Name"] for r in rows]
starts=[int(r["Start_Timestamp"]) for r in rows]; ends=[int(r["End_Timestamp"]) for r in rows]
import collections
print(len(rows), collections.Counter(n[:50] for n in names).most_common(14))
idx=[i for i,n in enumerate(names) if "premul" in n]
if len(idx) < 4: idx=[i for i,n in enumerate(names) if "peak_candidates" in n]
print("kernels per call:", idx[-1]-idx[-2])
for a,b in zip(idx[-4:-1], idx[-3:]):
    span=(ends[b-1]-starts[a])/1e3; busy=sum(ends[i]-starts[i] for i in range(a,b))/1e3
    print(f"call: span {span:.1f} us, kernel time {busy:.1f} us, kernels {b-a}")
a,b=idx[-2],idx[-1]
for i in range(a,b): print(f"  {(starts[i]-starts[a])/1e3:7.1f} +{(ends[i]-starts[i])/1e3:6.1f} us  {names[i][:70]}")
PY
