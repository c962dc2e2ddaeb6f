#!/bin/bash
# usage: scripts/prof_stats.sh <tag> [bench args...]   -- rocprofv3 kernel stats of bench.py into gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-diagnostics "$@" > gpurun_out/prof_$tag.log 2>&1
cp $(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats.csv; rm -rf gpurun_out/prof_$tag
python3 - <<PY
import csv,glob
f="gpurun_out/${tag}_kernel_stats.csv"
for r in list(csv.DictReader(open(f)))[:24]:
    print(r["Name"][:100], r["Calls"], r["AverageNs"][:8], r["Percentage"])
PY
tail -c 400 gpurun_out/prof_$tag.log | grep -o '"ms_per_step": [0-9.]*'
