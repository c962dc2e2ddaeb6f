#!/bin/bash
# usage (GPU box): bash scripts/probes/run_pkfma_probe.sh  -- the micro-probe alone, then beside a second process that keeps the GPU busy
cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O2 -o /tmp/pkfma_src1 scripts/probes/pkfma_src1.hip 2>/dev/null || exit 1
echo "alone:"; /tmp/pkfma_src1 5
python3 - <<'PY' &
# the co-runner of tests/test_gpu_colocation.py: full-size Code2Wav windows in a loop
import sys, time, threading
sys.path.insert(0, ".")
import torch
from ht_vllm_omni_amd.code2wav import Code2WavDecoder
from tests.codec_util import FULL_CODEC, make_codec_state
dec = Code2WavDecoder(FULL_CODEC, make_codec_state(FULL_CODEC, 0, device="cuda"))
codes = torch.randint(0, 2048, (1, 16, 50), device="cuda")
t0 = time.time()
while time.time() - t0 < 40:
    for _ in range(4):
        dec(codes)
    torch.cuda.synchronize()
PY
co=$!
sleep 20
echo "beside the Code2Wav loop of another process:"; /tmp/pkfma_src1 8
echo "beside it, second run:"; /tmp/pkfma_src1 5
wait $co
