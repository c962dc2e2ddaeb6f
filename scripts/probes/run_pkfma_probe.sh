#!/bin/bash
# usage (GPU box): bash scripts/probes/run_pkfma_probe.sh  -- the micro-probe alone, then beside a second process that keeps the GPU busy
cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O2 -o /tmp/pkfma_src1 scripts/probes/pkfma_src1.hip 2>/dev/null || exit 1
echo "alone:"; /tmp/pkfma_src1 5
python3 - <<'PY' &
import torch, time
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
t0 = time.time()
while time.time() - t0 < 25:
    for _ in range(50):
        b = a @ a
        c = torch.nn.functional.gelu(b)
    torch.cuda.synchronize()
PY
co=$!
sleep 8
echo "beside a GEMM / elementwise loop of another process:"; /tmp/pkfma_src1 8
echo "beside it, second run:"; /tmp/pkfma_src1 5
wait $co
