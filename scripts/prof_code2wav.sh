#!/bin/bash
# rocprofv3 kernel stats of the full-size Code2Wav decoder at a 325-frame window: scripts/prof_code2wav.sh <tag>
tag=${1:-c2w}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
rm -rf /tmp/prof_$tag
TORCH_LEG=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $GRAFT_REPO_ROOT/scripts/bench_code2wav.py ${2:-325} > $out/prof_$tag.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" $out/${tag}_kernel_stats.csv
head -30 $out/${tag}_kernel_stats.csv | cut -c1-200
