#!/bin/bash
# A/B two builds of the library on ONE box: ht_vllm_omni_amd/libomni_talker_{old,new}.so (boxes differ by ~1 %)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in old new; do
  cp ht_vllm_omni_amd/libomni_talker_$v.so ht_vllm_omni_amd/libomni_talker.so
  python bench.py --steps 64 --warmup 4 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$v', '%.4f' % j['ms_per_step'], '%.4f' % j['roofline']['breakdown']['backbone_ms'])"
done; done
