#!/bin/bash
# usage: scripts/prof_stats.sh <tag> [bench args...]   -- rocprofv3 kernel stats of bench.py into gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o $tag -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$tag/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(r["Name"][:100], r["Calls"], r["AverageNs"][:8], r["Percentage"])
PY
tail -c 400 gpurun_out/prof_$tag.log | grep -o '"ms_per_step": [0-9.]*'
