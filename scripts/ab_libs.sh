#!/bin/bash
# Round 6: A/B library builds on ONE box (boxes differ by ~1 %): scripts/ab_libs.sh <reps> <libA.so> <libB.so> ... [-- bench args]
# each lib is copied over ht_vllm_omni_amd/libomni_talker.so in turn (stale check off), bench.py --steps 64 printed per run
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
reps=$1; shift
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ "$1" == "--" ] && shift
cp ht_vllm_omni_amd/libomni_talker.so /tmp/libomni_talker_keep.so
for v in "${libs[@]}"; do cp "ht_vllm_omni_amd/$v" "/tmp/ab_$v"; done      # (one of them may BE libomni_talker.so)
for rep in $(seq 1 "$reps"); do for v in "${libs[@]}"; do
  cp "/tmp/ab_$v" ht_vllm_omni_amd/libomni_talker.so
  OMNI_SKIP_STALE_CHECK=1 python bench.py --steps 64 --warmup 4 --no-cpu-baseline --no-engine-loop "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); f=j['roofline']['families']
print('$v', 'step %.4f' % j['ms_per_step'], 'cp %.4f' % f['code_predictor_phase_ms'], 'bb_chain %.4f' % f['backbone_chain_ms'], 'attn %.4f' % f['backbone_attention_ms'], 'head %.4f' % f['lm_head_sampler_ms'], 'ttfa %.2f' % j.get('p50_ttfa_ms', 0))"
done; done
cp /tmp/libomni_talker_keep.so ht_vllm_omni_amd/libomni_talker.so
