#!/bin/bash
# Round-3 A/B table of the persistent-chain arms on the W3 step (one process, one hipGraph per setting, round-robin timing)
# and the in-kernel timelines of the default build.  usage (GPU box): bash scripts/r03_ab_table.sh > gpurun_out/r03_ab_table.txt
export OMNI_TALKER_DEBUG=1
python scripts/ab_knobs.py --steps 24 --rounds 3 \
  cp_chain=0,bb_chain=0,bb_engine=0,bb_pp=0 \
  cp_chain=1,bb_chain=0,bb_engine=0,bb_pp=0 \
  cp_chain=1,bb_chain=1,bb_engine=0,bb_pp=0 \
  cp_chain=1,bb_chain=2,bb_engine=0,bb_pp=0 \
  cp_chain=1,bb_chain=1,bb_engine=1,bb_pp=0 \
  cp_chain=1,bb_chain=1,bb_engine=0,bb_pp=1 \
  cp_chain=1,bb_chain=1,bb_engine=0,bb_pp=0,bb_xw=1 \
  cp_chain=1,bb_chain=1,bb_engine=0,bb_pp=0,bb_xw=0,bb_deep=1 \
  cp_chain=1,bb_chain=1,bb_engine=0,bb_pp=0,bb_xw=0,bb_deep=0,chain_mode=7:1:16 \
  cp_chain=1,bb_chain=1,bb_engine=0,bb_pp=0,bb_xw=0,bb_deep=0,chain_mode=7:1:0 \
  cp_chain=1,bb_chain=1,bb_engine=0,bb_pp=0,chain_mode=7:1:4 \
  cp_chain=1,bb_chain=1,bb_engine=0,bb_pp=0,chain_mode=8:1:1 2>&1 | grep -v "weights generated"
echo "---- code-predictor chain timeline (default build)"
python scripts/chain_timeline.py 2>&1 | tail -40
echo "---- backbone chain timeline (default build)"
BB_STAMPS=bb python scripts/bb_timeline.py 2>&1 | tail -8
echo "---- partly filled batches: launch path vs chain (code-predictor half = step - backbone)"
for b in 1 16 32 48; do echo "B=$b"; python scripts/ab_knobs.py --batch $b --steps 24 --rounds 2 cp_chain=0 cp_chain=1 2>&1 | tail -2; done
echo "---- two-group backbone chain timeline (bb_pp=1)"
BB_STAMPS=pp python scripts/bb_timeline.py 2>&1 | tail -8
