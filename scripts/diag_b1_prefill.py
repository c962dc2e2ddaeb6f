"""Where do the first-request milliseconds after the prefill go at B = 1? (diagnostic): statement-level timing of bench.setup_requests' tail."""
import os, sys, time, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from ht_vllm_omni_amd import ops
args = argparse.Namespace(model="tts-1.7b", kv="fp8", num_blocks=8192, batch=1, sub_batches=1, warmup=8, steps=32, ttfa_steps=2, device_weights=True, ctx_extra=0, tp_force=False, parallel="tp", target_ctx=352, allreduce="oneshot", prefill_gemm=sys.argv[1] if len(sys.argv) > 1 else "tile")
d, w, eng = bench.build_engine(args, 0, 1)
eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42)
T = 153
x = (torch.randn(T, d.hidden) * 0.05).to(torch.bfloat16).cuda()
pos = torch.arange(T, dtype=torch.int32).cuda(); req = torch.zeros(T, dtype=torch.int32).cuda()
eng.block_table[0, :16] = torch.arange(1, 17, dtype=torch.int32)
slots = (16 + torch.arange(T)).cuda()
def t(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) * 1e3:8.3f} ms"); return r
for rep in range(2):
    print("--- pass", rep)
    hid = t("eng.prefill", lambda: eng.prefill(x, pos, req, slots))
    last = t("last index .cuda()", lambda: torch.tensor(np.array([T - 1])).cuda())
    hl = t("hid[last]", lambda: hid[last])
    logits = t("compute_logits", lambda: eng.compute_logits(hl))
    t("seen.zero_ + set", lambda: (eng.seen.zero_(), eng.seen[:1, d.codec_pad_id].fill_(1), eng.steps.zero_()))
    s = eng.sampling
    t("ops.sample", lambda: ops.sample(logits, greedy=bool(s["greedy"]), temperature=s["temperature"], top_k=s["top_k"], rep_penalty=s["rep_penalty"], seen=eng.seen[:1], seed=s["seed"], steps=eng.steps[:1], inc_steps=True))
