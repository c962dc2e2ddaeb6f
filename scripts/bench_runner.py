"""Wall time per engine step THROUGH the worker / runner boundary (execute_model + sample_tokens with its host copy and
bookkeeping), W3 shape, B = 64 -- next to bench.py's bare hipGraph replay (diagnostic)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.payloads import (OmniCachedRequestData, OmniNewRequestData, OmniSchedulerOutput, SamplingParams, encode_tensor)
from ht_vllm_omni_amd.sched import BlockPool
from ht_vllm_omni_amd.weights import make_weights
from ht_vllm_omni_amd.worker import MI355XARWorker, make_config

d = get_dims(os.environ.get("MODEL", "tts-1.7b"))
w = make_weights(d, seed=1234, std=0.02)
B, bs, nb = 64, 16, 8192
sp = SamplingParams(temperature=0.9, top_k=50, repetition_penalty=1.05, seed=42, max_tokens=100000, stop_token_ids=())
cfg = make_config(d, kv_cache_dtype="fp8", block_size=bs, max_num_seqs=B, num_gpu_blocks_override=nb, weights=w, default_sampling_params=sp)
wk = MI355XARWorker(cfg, local_rank=0, rank=0)
wk.init_device(); wk.load_model(); wk.initialize_from_config(None)
wk.engine.set_sampling(cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
wk.compile_or_warm_up_model()
run = wk.model_runner
pool = BlockPool(nb, bs)
g = torch.Generator().manual_seed(7)
lens = np.random.default_rng(7).integers(32, 161, size=B).tolist()
new = []
for r, n in enumerate(lens):
    pool.allocate(f"r{r}", n + 400)
    info = {"talker_prompt_embeds": encode_tensor((torch.randn(n, d.hidden, generator=g) * 0.05).to(torch.bfloat16)),
            "tts_pad_embed": encode_tensor((torch.randn(d.hidden, generator=g) * 0.05).to(torch.bfloat16)),
            "tailing_text_hidden": encode_tensor((torch.randn(20, d.hidden, generator=g) * 0.05).to(torch.bfloat16))}
    new.append(OmniNewRequestData(req_id=f"r{r}", prompt_token_ids=[d.codec_pad_id] * n, block_ids=(pool.block_ids(f"r{r}"),),
                                  sampling_params=sp, additional_information=info))
so = OmniSchedulerOutput(scheduled_new_reqs=new, num_scheduled_tokens={f"r{r}": n for r, n in enumerate(lens)}, total_num_scheduled_tokens=sum(lens))
wk.execute_model(so); wk.sample_tokens(None)
keys = [f"r{r}" for r in range(B)]
def step():
    so = OmniSchedulerOutput(scheduled_cached_reqs=OmniCachedRequestData(req_ids=keys, new_block_ids=[None] * B),
                             num_scheduled_tokens={k: 1 for k in keys}, total_num_scheduled_tokens=B)
    t0 = time.perf_counter(); wk.execute_model(so); t1 = time.perf_counter(); out = wk.sample_tokens(None); t2 = time.perf_counter()
    return t1 - t0, t2 - t1
for _ in range(10): step()
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter(); ex = sa = 0.0
for _ in range(N):
    a, b = step(); ex += a; sa += b
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print(f"runner wall per step {dt*1e3:.3f} ms  ({B/dt:.0f} tok/s)  execute_model {ex/N*1e3:.3f} ms  sample_tokens {sa/N*1e3:.3f} ms  replays {run.cudagraph_stats}")

# ---- the same through the scheduler (engine-core loop: schedule -> execute_model -> sample_tokens -> update_from_output)
from ht_vllm_omni_amd.scheduler import MI355XARScheduler, Request, TalkerStageEngine
wk.shutdown()
wk = MI355XARWorker(cfg, local_rank=0, rank=0)
wk.init_device(); wk.load_model(); wk.initialize_from_config(None)
wk.engine.set_sampling(cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
wk.compile_or_warm_up_model()
sched = MI355XARScheduler(num_blocks=nb, block_size=bs, max_num_seqs=B, max_num_batched_tokens=8192, max_model_len=d.max_model_len,
                          need_send_cache=False)
core = TalkerStageEngine(wk, sched)
sp2 = SamplingParams(temperature=0.9, top_k=50, repetition_penalty=1.05, seed=42, max_tokens=400, stop_token_ids=())
for r, n in enumerate(lens):
    info = {"talker_prompt_embeds": encode_tensor((torch.randn(n, d.hidden, generator=g) * 0.05).to(torch.bfloat16)),
            "tts_pad_embed": encode_tensor((torch.randn(d.hidden, generator=g) * 0.05).to(torch.bfloat16))}
    core.add_request(Request(request_id=f"s{r}", num_prompt_tokens=n, prompt_token_ids=[d.codec_pad_id] * n, sampling_params=sp2,
                             additional_information=info, ignore_eos=True))
for _ in range(12): core.step()
torch.cuda.synchronize()
t0 = time.perf_counter(); ts = tu = 0.0
for _ in range(N):
    a = time.perf_counter(); so = sched.schedule(); b = time.perf_counter()
    first = wk.execute_model(so); out = first if first is not None else wk.sample_tokens(None)
    c = time.perf_counter(); sched.update_from_output(so, out); tu += time.perf_counter() - c; ts += b - a
dt = (time.perf_counter() - t0) / N
print(f"engine-core loop wall per step {dt*1e3:.3f} ms  ({B/dt:.0f} tok/s)  schedule {ts/N*1e3:.3f} ms  update_from_output {tu/N*1e3:.3f} ms")
