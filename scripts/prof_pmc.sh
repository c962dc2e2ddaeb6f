#!/bin/bash
# usage: scripts/prof_pmc.sh <tag>  -- FETCH_SIZE and WRITE_SIZE passes (separate runs) of a 3-step bench AT THE HEADLINE CONTEXT.
# rocprofv3's counter collection segfaults inside hipGraphLaunch after a few hundred replays on this stack, so the batch is
# not advanced to mean ctx 352 by ~250 untimed steps as in the timed bench: the decode starts 250 positions later instead
# (--target-ctx 0 --ctx-extra 250: the attention reads the same NUMBER of KV bytes; those rows were never written -- byte
# traffic does not depend on the values).  Writes gpurun_out/pmc_<tag>_step_traffic.json with the mean context of the measured
# step, which bench.py reads back as roofline.traffic only when its own context matches
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_${tag}_$c -o p -- python3 bench.py --steps 3 --warmup 1 --ttfa-steps 3 --target-ctx 0 --ctx-extra 250 --no-cpu-baseline --no-diagnostics --device-weights > gpurun_out/pmc_${tag}_$c.json 2> gpurun_out/pmc_${tag}_$c.log
done
f=$(find gpurun_out/pmc_${tag}_FETCH_SIZE -name '*counter_collection.csv' | head -1)
w=$(find gpurun_out/pmc_${tag}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
ctx=$(python3 -c "import json,sys; print(json.loads(open('gpurun_out/pmc_${tag}_FETCH_SIZE.json').read().strip().splitlines()[-1])['config']['mean_ctx'])")
python3 scripts/pmc_step_traffic.py "$f" "$w" gpurun_out/pmc_${tag}_step_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the last decode step of bench.py --steps 3 --warmup 1 --ttfa-steps 3 --target-ctx 0 --ctx-extra 250 --no-cpu-baseline --device-weights; FETCH_SIZE doubled per MI355X_MICROARCH.md" "$ctx"
rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE
