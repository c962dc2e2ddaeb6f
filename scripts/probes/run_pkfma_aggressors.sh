#!/bin/bash
# usage (GPU box): bash scripts/probes/run_pkfma_aggressors.sh -- the micro-probe beside a second process looping ONE feature (aggressor.hip)
cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O2 -o /tmp/pkfma_src1 scripts/probes/pkfma_src1.hip 2>/dev/null || exit 1
hipcc --offload-arch=gfx950 -O2 -o /tmp/aggressor scripts/probes/aggressor.hip 2>/dev/null || exit 1
for mode in ${MODES:-ldsdma mfma setprio lds vmem}; do
  /tmp/aggressor $mode 14 > /tmp/agg_$mode.txt 2>&1 &
  co=$!
  sleep 4
  echo "== beside: $mode"; /tmp/pkfma_src1 6 
  wait $co; cat /tmp/agg_$mode.txt
done
