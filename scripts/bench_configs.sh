#!/bin/bash
# other configurations of BASELINE.json / SURVEY 8d next to the headline W3 line (one JSON line each into gpurun_out/)
cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift; python bench.py --no-cpu-baseline --no-engine-loop "$@" 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; b=r.get('breakdown') or {}
print('$tag', 'ms/step %.3f' % j['ms_per_step'], 'tok/s %.0f' % j['value'], 'frac %.3f' % r['frac'], 'ctx %.0f' % j['config']['mean_ctx'], 'ttfa %.1f' % j['p50_ttfa_ms'], 'bb_ms %.3f bb_frac %.3f' % (b.get('backbone_ms', 0), b.get('backbone_frac', 0)))"; }
run W3-fp8 --steps 128
run W3-bf16kv --kv bf16 --steps 128
run W3-int8kv --kv int8 --steps 128
run W2-0.6b-bf16kv --model tts-0.6b --kv bf16 --steps 128
run W2-0.6b-b16 --model tts-0.6b --kv bf16 --batch 16 --steps 128
run W2-0.6b-b1 --model tts-0.6b --kv bf16 --batch 1 --steps 128 --ttfa-steps 2
run W3-b1 --batch 1 --steps 128 --ttfa-steps 2
run W3-b40 --batch 40 --steps 128
run W3-b48 --batch 48 --steps 128
run W3-tp-force --tp-force --steps 128
run W3-fp16kv --kv fp16 --steps 128
run W3-ctx2048 --steps 64 --ctx-extra 1900 --num-blocks 12000 --target-ctx 0
run W3-ctx4000 --steps 32 --ctx-extra 3800 --num-blocks 20000 --target-ctx 0
run omni-talker-int8 --model omni-talker --kv int8 --steps 128
run W3-b128-2chains --batch 128 --sub-batches 2 --steps 64 --num-blocks 16384
run W3-b256-4chains --batch 256 --sub-batches 4 --steps 64 --num-blocks 32768
