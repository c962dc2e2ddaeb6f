#!/bin/bash
# usage: scripts/prof_pmc.sh <tag>  -- FETCH_SIZE and WRITE_SIZE passes (separate runs) of a 3-step bench
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -o p -- python3 bench.py --steps 3 --warmup 1 --ttfa-steps 3 --no-cpu-baseline --device-weights > gpurun_out/pmc_${tag}_$c.log 2>&1
done
f=$(find gpurun_out/pmc_${tag}_FETCH_SIZE -name '*counter_collection.csv' | head -1)
w=$(find gpurun_out/pmc_${tag}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 scripts/pmc_step_traffic.py "$f" "$w" gpurun_out/pmc_${tag}_step_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the last decode step of bench.py --steps 3 --warmup 1 --ttfa-steps 3 --no-cpu-baseline --device-weights; FETCH_SIZE doubled per MI355X_MICROARCH.md; mean ctx ~105"
rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE
