#!/bin/bash
# usage (GPU box): bash scripts/probes/run_pkfma_neighbours.sh  -- which neighbour makes op_sel[1] = 1 fail?  The micro-probe beside a second
# process that loops ONE kind of kernel: the hand-written prefill GEMM (LDS-DMA, s_setprio), the decode attention, a torch matmul.
cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O2 -o /tmp/pkfma_src1 scripts/probes/pkfma_src1.hip 2>/dev/null || exit 1
neighbour() {
python3 - "$1" <<'PY' &
import sys, time
sys.path.insert(0, ".")
import torch
from ht_vllm_omni_amd import ops, _lib as L
from ht_vllm_omni_amd.engine import frag_shuffle
kind = sys.argv[1]
g = torch.Generator().manual_seed(1)
if kind == "gemm_tile":
    x = torch.randn(4096, 2048, generator=g).to(torch.bfloat16).cuda()
    w = frag_shuffle((torch.randn(2048, 2048, generator=g) * 0.03).to(torch.bfloat16)).cuda()
    run = lambda: ops.gemm_tile(x, w)
elif kind == "attention":
    B, hq, hkv, D, bs, ctx = 64, 16, 8, 128, 16, 356
    nb = B * (ctx // bs + 2) + 1
    cache = torch.randint(0, 120, (2, nb, bs, hkv, D), generator=g, dtype=torch.uint8).cuda()
    per = ctx // bs + 2
    bt = (torch.arange(B * per, dtype=torch.int32).view(B, per) + 1).cuda()
    seq = torch.full((B,), ctx, dtype=torch.int32).cuda()
    q = torch.randn(B, hq * D, generator=g).to(torch.bfloat16).cuda()
    run = lambda: ops.paged_attn_decode(q, cache[0], cache[1], bt, seq, q_heads=hq, kv_heads=hkv, head_dim=D, block_size=bs,
                                        kv_dtype=L.KV_CODES["fp8"], k_scale=0.5, v_scale=1.0, max_seq_len=1024, split=False)
else:
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
    run = lambda: a @ a
run(); torch.cuda.synchronize()
t0 = time.time()
while time.time() - t0 < 22:
    for _ in range(50):
        run()
    torch.cuda.synchronize()
PY
}
for kind in gemm_tile attention matmul; do
  neighbour $kind; co=$!
  sleep 12
  echo "== beside a loop of: $kind"; /tmp/pkfma_src1 6 | grep "launches\|src1 high"
  wait $co
done
