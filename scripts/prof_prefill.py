"""Where does the prefill time go? (diagnostic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, argparse
import bench
args = argparse.Namespace(model="tts-1.7b", kv="fp8", num_blocks=8192, batch=64, sub_batches=1, warmup=8, steps=64, ttfa_steps=16, device_weights=True, ctx_extra=0, tp_force=False, parallel="tp", target_ctx=0, allreduce="oneshot", prefill_gemm=os.environ.get("PREFILL_GEMM", "tile"))
d, w, eng = bench.build_engine(args, 0, 1)
eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42)
for i in range(3):
    lens, ms = bench.setup_requests(d, eng, args)
    print("prefill ms", ms)
