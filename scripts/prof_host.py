"""cProfile of the engine-core loop's host side (diagnostic): 150 steady-state steps, W3 shape, B = 64."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.payloads import SamplingParams, encode_tensor
from ht_vllm_omni_amd.scheduler import MI355XARScheduler, Request, TalkerStageEngine
from ht_vllm_omni_amd.weights import make_weights
from ht_vllm_omni_amd.worker import MI355XARWorker, make_config
d = get_dims("tts-1.7b").with_(layers=2)          # host work does not depend on the layer count
w = make_weights(d, seed=1234, std=0.02)
B, bs, nb = 64, 16, 8192
sp = SamplingParams(temperature=0.9, top_k=50, repetition_penalty=1.05, seed=42, max_tokens=400, stop_token_ids=())
cfg = make_config(d, kv_cache_dtype="fp8", block_size=bs, max_num_seqs=B, num_gpu_blocks_override=nb, weights=w, default_sampling_params=sp)
wk = MI355XARWorker(cfg, local_rank=0, rank=0)
wk.init_device(); wk.load_model(); wk.initialize_from_config(None)
wk.engine.set_sampling(cp_greedy=0, cp_temperature=0.9, cp_top_k=50)
wk.compile_or_warm_up_model()
sched = MI355XARScheduler(num_blocks=nb, block_size=bs, max_num_seqs=B, max_num_batched_tokens=8192, max_model_len=d.max_model_len, need_send_cache=False)
core = TalkerStageEngine(wk, sched)
g = torch.Generator().manual_seed(7)
for r, n in enumerate(np.random.default_rng(7).integers(32, 161, size=B).tolist()):
    info = {"talker_prompt_embeds": encode_tensor((torch.randn(n, d.hidden, generator=g) * 0.05).to(torch.bfloat16)),
            "tts_pad_embed": encode_tensor((torch.randn(d.hidden, generator=g) * 0.05).to(torch.bfloat16))}
    core.add_request(Request(request_id=f"s{r}", num_prompt_tokens=n, prompt_token_ids=[d.codec_pad_id] * n, sampling_params=sp,
                             additional_information=info, ignore_eos=True))
for _ in range(12): core.step()
pr = cProfile.Profile(); pr.enable()
for _ in range(150): core.step()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:4000])
