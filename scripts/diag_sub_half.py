"""diagnostic: two 64-row ranges on half-grid backbone chains at the bench shape -- which stage's wait runs out, eager vs graph"""
import os, sys, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 28
args = types.SimpleNamespace(allreduce="rccl", model="tts-1.7b", kv="fp8", batch=128, num_blocks=8192, device_weights=True, parallel="tp", sub_batches=2,
                             tp_force=False, prefill_gemm="tile", warmup=0, steps=64, ttfa_steps=0, ctx_extra=0, target_ctx=-1)
torch.cuda.set_device(0)
import ht_vllm_omni_amd.config as cfg
if layers != 28:
    _g = cfg.get_dims
    cfg.get_dims = lambda m: _g(m).with_(layers=layers)
d, w, eng = bench.build_engine(args, 0, 1)
eng.set_sampling(greedy=0, temperature=0.9, top_k=50, rep_penalty=1.05, seed=42, cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
lens, _ = bench.setup_requests(d, eng, args)
B = 128
for i in range(6):
    t0 = time.time()
    eng.decode_step(B); torch.cuda.synchronize()
    print("eager step", i, "ms", round((time.time() - t0) * 1e3, 2), "ran", eng.chains_ran(), "err", hex(eng.chain_error()), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    eng.decode_step(B)
for i in range(6):
    t0 = time.time()
    g.replay(); torch.cuda.synchronize()
    print("graph step", i, "ms", round((time.time() - t0) * 1e3, 2), "err", hex(eng.chain_error()), flush=True)
