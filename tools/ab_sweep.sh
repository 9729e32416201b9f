#!/bin/bash
# tools/plan_ab.py over a list of shapes: tools/ab_sweep.sh <out file> <ENV> "<values>" "<log2n batch dtype flags>" ...
out=$1; env=$2; vals=$3; shift 3
: > "$out"
for c in "$@"; do
  set -- $c
  timeout -k 10 200 python3 "$(dirname "$0")/plan_ab.py" --log2n $1 --batch $2 --dtype $3 --flags $4 --env $env --values $vals 2>&1 | grep -v amdgpu.ids >> "$out" || exit 1
done
