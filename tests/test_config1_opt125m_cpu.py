"""BASELINE config #1 -- "OPT-125m text-only greedy decode via the engine on CPU (AR plumbing, no GPU)" (SURVEY 8d W1).

Nothing of OPT is in the reference tree or in the product: this is the PLUMBING of the AR stage -- MI355XARScheduler
(admission under a token budget, chunked prefill, block allocation, stop conditions) driving MI355XARModelRunner's
two-phase contract -- run on a CPU stand-in engine whose arithmetic is HF transformers' OPTForCausalLM at the OPT-125m
shape (random-init weights from the config: no checkpoint, no network).  The stand-in lives in tests/ (like tests/fakes.py):
the product has no CPU path.  Oracle = HF `generate(do_sample=False)` on the same module: the engine loop must emit exactly
its tokens for 4 concurrent requests x 64 new tokens, through chunked prefill and staggered arrival.
"""
import pytest
import torch

from ht_vllm_omni_amd.config import TalkerDims
from ht_vllm_omni_amd.payloads import SamplingParams, encode_tensor
from ht_vllm_omni_amd.runner import MI355XARModelRunner
from ht_vllm_omni_amd.scheduler import MI355XARScheduler, Request, TalkerStageEngine

BF16 = torch.bfloat16


class OptCpuEngine:
    """The engine surface the runner drives (persistent per-row buffers, prefill / compute_logits / sample_rows /
    decode_step), text-only: hidden states come from an HF OPT decoder with one DynamicCache per batch row; there are no
    audio codes (Q = 1) and the text-step queue is unused."""

    def __init__(self, model, dims, max_batch=4, block_size=16, num_blocks=64):
        from transformers.cache_utils import DynamicCache
        self._Cache = DynamicCache
        self.m, self.d = model, dims
        self.max_batch, self.block_size, self.num_blocks = max_batch, block_size, num_blocks
        self.kv_dtype, self.bt_stride = "fp32", dims.max_model_len // block_size
        z = torch.zeros
        self.input_ids = z(max_batch, dtype=torch.int32)
        self.positions = z(max_batch, dtype=torch.int32)
        self.seq_lens = z(max_batch, dtype=torch.int32)
        self.block_table = z(max_batch, self.bt_stride, dtype=torch.int32)
        self.slot_mapping = z(max_batch, dtype=torch.int64)
        self.last_hidden = z(max_batch, dims.hidden, dtype=BF16)
        self.text_step = z(max_batch, dims.hidden, dtype=BF16)
        self.inputs_embeds = z(max_batch, dims.hidden, dtype=BF16)
        self.audio_codes = z(max_batch, dims.num_code_groups, dtype=torch.int64)
        self.seen = z(max_batch, dims.vocab, dtype=torch.uint8)
        self.steps = z(max_batch, dtype=torch.int32)
        for n in ("greedy", "top_k", "seed"):
            setattr(self, "row_" + n, z(max_batch, dtype=torch.int32))
        for n in ("temperature", "top_p", "rep_penalty"):
            setattr(self, "row_" + n, torch.ones(max_batch))
        self.num_live = torch.full((1,), max_batch, dtype=torch.int32)
        self.kv_caches = []
        self.caches: dict[int, object] = {}          # block id of the row's first block -> HF cache (the row's identity)
        self._hid32: dict[int, torch.Tensor] = {}

    def set_row_sampling(self, row, **kw):
        assert kw["greedy"], "config #1 is greedy"

    def _key(self, row):
        return int(self.block_table[row, 0])

    @torch.inference_mode()
    def prefill(self, x, positions, req_of_tok, slot_mapping, block_table=None):
        out = torch.empty(x.shape[0], self.d.hidden)
        for r in sorted(set(req_of_tok.tolist())):
            idx = (req_of_tok == r).nonzero().flatten()
            p0 = int(positions[idx[0]])
            key = self._key(r)
            if p0 == 0:
                self.caches[key] = self._Cache(config=self.m.config)
            cache = self.caches[key]
            ids = self._ids[key][p0:p0 + len(idx)]
            o = self.m.model.decoder(input_ids=ids[None], past_key_values=cache, use_cache=True,
                                     attention_mask=torch.ones(1, p0 + len(idx), dtype=torch.long))
            out[idx] = o.last_hidden_state[0]
        self._hid_rows = out
        return out.to(BF16)

    def compute_logits(self, hidden, round_bf16=True):
        # the runner hands back bf16 rows of what prefill returned: use the exact fp32 rows of the same positions
        rows = [int((self._hid_rows.to(BF16) == h).all(1).nonzero()[0]) for h in hidden]
        return self.m.lm_head(self._hid_rows[rows])

    def sample_rows(self, logits, rows, *, seen=None, steps=None):
        if steps is not None:
            steps += 1
        return logits.argmax(-1).to(torch.int32)

    @torch.inference_mode()
    def decode_step(self, B, advance=True):
        B = min(B, int(self.num_live))
        for r in range(B):
            cache = self.caches[self._key(r)]
            pos = int(self.positions[r])
            o = self.m.model.decoder(input_ids=self.input_ids[r].long().reshape(1, 1), past_key_values=cache, use_cache=True,
                                     attention_mask=torch.ones(1, pos + 1, dtype=torch.long))
            h = o.last_hidden_state[0, -1]
            self.last_hidden[r] = h.to(BF16)
            self.input_ids[r] = int(self.m.lm_head(h).argmax())
            self.slot_mapping[r] = int(self.block_table[r, pos // self.block_size]) * self.block_size + pos % self.block_size
        self.steps[:B] += 1
        if advance:
            self.positions[:B] += 1
            self.seq_lens[:B] += 1


class _Worker:
    def __init__(self, runner):
        self.model_runner = runner

    def execute_model(self, so):
        return self.model_runner.execute_model(so)

    def sample_tokens(self, g):
        return self.model_runner.sample_tokens(g)


@pytest.mark.timeout(600)
def test_opt125m_shape_greedy_decode_through_scheduler_and_runner_on_cpu():
    from transformers import OPTConfig, OPTForCausalLM
    torch.manual_seed(0)
    cfg = OPTConfig(vocab_size=50272, hidden_size=768, num_hidden_layers=12, ffn_dim=3072, num_attention_heads=12,
                    max_position_embeddings=2048, word_embed_proj_dim=768, do_layer_norm_before=True)     # facebook/opt-125m
    model = OPTForCausalLM(cfg).eval()
    dims = TalkerDims(name="opt-125m", hidden=768, layers=12, q_heads=12, kv_heads=12, head_dim=64, inter=3072, vocab=50272,
                      codebook=50272, eos_id=2, codec_pad_id=1, num_code_groups=1, rope_theta=1.0, eps=1e-5, cp_hidden=768,
                      cp_layers=0, cp_q_heads=1, cp_kv_heads=1, cp_head_dim=64, cp_inter=64, cp_rope_theta=1.0, max_model_len=512)
    eng = OptCpuEngine(model, dims, max_batch=4)
    run = MI355XARModelRunner(eng, use_graphs=False, engine_output_type="text")
    sched = MI355XARScheduler(num_blocks=64, block_size=16, max_num_seqs=4, max_num_batched_tokens=24, max_model_len=512)
    core = TalkerStageEngine(_Worker(run), sched)
    g = torch.Generator().manual_seed(1)
    prompts = {f"p{i}": torch.randint(3, 50000, (n,), generator=g) for i, n in enumerate((7, 30, 12, 19))}   # 30 > budget: chunked
    eng._ids = {}
    n_new = 64
    want = {}
    with torch.inference_mode():
        for k, ids in prompts.items():
            out = model.generate(ids[None], max_new_tokens=n_new, do_sample=False, eos_token_id=None, pad_token_id=1)
            want[k] = out[0, len(ids):].tolist()
    reqs = []
    for k, ids in prompts.items():
        sp = SamplingParams(temperature=0.0, top_k=0, repetition_penalty=1.0, max_tokens=n_new, stop_token_ids=())
        # the runner's prefill input is an embedding matrix: the stand-in looks the ids up again by row identity; tts_pad: unused
        info = {"talker_prompt_embeds": encode_tensor(torch.zeros(len(ids), dims.hidden, dtype=BF16)),
                "tts_pad_embed": encode_tensor(torch.zeros(dims.hidden, dtype=BF16))}
        reqs.append(Request(request_id=k, num_prompt_tokens=len(ids), prompt_token_ids=ids.tolist(), sampling_params=sp,
                            additional_information=info, ignore_eos=True))
    tokens = {k: [] for k in prompts}
    pending = list(reqs)
    for step in range(400):
        if pending and step % 3 == 0:                      # staggered arrival
            r = pending.pop(0)
            core.add_request(r)
        so = sched.schedule()              # the stand-in identifies a row by its first block id: register prompts on admission
        for nr in so.scheduled_new_reqs:
            eng._ids[nr.block_ids[0][0]] = prompts[nr.req_id]
        if so.total_num_scheduled_tokens or so.finished_req_ids:
            first = run.execute_model(so)
            out = first if first is not None else run.sample_tokens(None)
            for o in sched.update_from_output(so, out):
                tokens[o.request_id] += o.new_token_ids
        if not pending and not sched.has_unfinished_requests():
            break
    run.execute_model(sched.schedule())      # delivers the last finished ids: the runner drops its rows
    assert {k: len(v) for k, v in tokens.items()} == {k: n_new for k in prompts}
    for k in prompts:
        assert tokens[k] == want[k], f"{k}: first divergence at {next(i for i, (a, b) in enumerate(zip(tokens[k], want[k])) if a != b)}"
    assert sched.pool.num_free == 63 and run.rows == []
