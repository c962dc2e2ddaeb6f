"""CPU stand-in for TalkerEngine so the runner's host logic is testable without a GPU (the reference tests its
runner the same way: object.__new__ + dummy buffers + a fake talker_mtp, T/worker/test_omni_gpu_model_runner.py:41-92)."""
import torch

BF16 = torch.bfloat16


class FakeEngine:
    def __init__(self, dims, max_batch=8, block_size=16, num_blocks=64):
        d = self.d = dims
        self.max_batch, self.block_size, self.num_blocks = max_batch, block_size, num_blocks
        self.kv_dtype = "bf16"
        self.bt_stride = d.max_model_len // block_size
        z = torch.zeros
        self.ids_status = z(max_batch + 4, dtype=torch.int32)       # ids + the step's status words (engine.TalkerEngine)
        self.input_ids = self.ids_status[:max_batch]
        self.status = self.ids_status[max_batch:]
        self.fault_next = False            # the next decode step "times out" in a persistent chain: garbage + status word 0
        self.persistent_chains = True
        self.positions = z(max_batch, dtype=torch.int32)
        self.seq_lens = z(max_batch, dtype=torch.int32)
        self.block_table = z(max_batch, self.bt_stride, dtype=torch.int32)
        self.slot_mapping = z(max_batch, dtype=torch.int64)
        self.last_hidden = z(max_batch, d.hidden, dtype=BF16)
        self.text_step = z(max_batch, d.hidden, dtype=BF16)
        self.inputs_embeds = z(max_batch, d.hidden, dtype=BF16)
        self.audio_codes = z(max_batch, d.num_code_groups, dtype=torch.int64)
        self.logits = z(max_batch, d.vocab)
        self.seen = z(max_batch, d.vocab, dtype=torch.uint8)
        self.steps = z(max_batch, dtype=torch.int32)
        self.row_greedy = torch.ones(max_batch, dtype=torch.int32)
        self.row_temperature = torch.ones(max_batch)
        self.row_top_k = z(max_batch, dtype=torch.int32)
        self.row_top_p = torch.ones(max_batch)
        self.row_rep_penalty = torch.ones(max_batch)
        self.row_seed = z(max_batch, dtype=torch.int32)
        self.num_live = torch.full((1,), max_batch, dtype=torch.int32)
        self.kv_caches = [torch.arange(2 * num_blocks * block_size * d.kv_heads * 4, dtype=torch.float32)
                          .reshape(2, num_blocks, block_size, d.kv_heads, 4) + 1000 * l for l in range(d.layers)]
        g = torch.Generator().manual_seed(99)
        self.embed = torch.randn(d.vocab, d.hidden, generator=g).to(BF16)
        self.cp_embed = torch.randn(d.num_code_groups - 1, d.codebook, d.hidden, generator=g).to(BF16)
        self.sampling = {}
        self.calls = []

    def set_sampling(self, **kw):
        self.sampling.update(kw)

    def set_row_sampling(self, row, *, greedy, temperature, top_k, top_p, rep_penalty, seed):
        if 0.0 < top_p < 1.0 and not 0 < top_k <= 1024:
            raise ValueError("top_p < 1 needs 0 < top_k <= 1024 on this path")
        self.row_greedy[row], self.row_temperature[row], self.row_top_k[row] = int(greedy), temperature or 1.0, top_k
        self.row_top_p[row], self.row_rep_penalty[row] = top_p, rep_penalty
        seed &= 0xFFFFFFFF
        self.row_seed[row] = seed - (1 << 32) if seed >= (1 << 31) else seed

    def sample_rows(self, logits, rows, *, seen=None, steps=None):
        self.calls.append(("sample_rows", rows.tolist(), self.row_seed[rows].tolist(), self.row_top_k[rows].tolist()))
        return self.sample(logits, greedy=True, seen=seen, steps=steps)

    def prefill(self, x, positions, req_of_tok, slot_mapping, block_table=None, rope_positions=None):
        self.calls.append(("prefill", x.shape[0], positions.tolist(), req_of_tok.tolist(), slot_mapping.tolist()))
        return (x.float() * 2).to(BF16)                       # "hidden" = 2 * embedding

    def compute_logits(self, hidden, round_bf16=True):
        lg = torch.full((hidden.shape[0], self.d.vocab), float("-inf"))
        for i in range(hidden.shape[0]):                       # deterministic "argmax" id from the hidden state
            lg[i, 1 + int(hidden[i, 0].float().abs().item() * 8) % (self.d.codebook - 1)] = 0.0
        return lg

    def sample(self, logits, *, greedy, temperature=1.0, top_k=0, top_p=1.0, rep_penalty=1.0, seen=None, seed=0, steps=None):
        ids = logits.argmax(-1).to(torch.int32)
        if seen is not None:
            seen[torch.arange(len(ids)), ids.long()] = 1
        if steps is not None:
            steps += 1
        return ids

    def set_chains(self, on):
        self.persistent_chains = bool(on)
        self.calls.append(("set_chains", bool(on)))

    def recover_from_chain_timeout(self):
        self.status.zero_()
        self.persistent_chains = False
        self.calls.append(("recover",))

    def decode_step(self, B, advance=True):
        B = min(B, int(self.num_live))          # the native step's contract: rows past the live count are inert
        self.calls.append(("decode", B, self.input_ids[:B].tolist(), self.positions[:B].tolist()))
        if self.fault_next and self.persistent_chains:
            # a timed-out chain: every output of the step is garbage, the state still advances, the status word says so
            self.fault_next = False
            self.status[0] = 0x1234
            self.input_ids[:B] = 1
            self.audio_codes[:B] = 7
            self.last_hidden[:B] = 99.0
            self.seen[:B, 1] = 1
            self.steps[:B] += 1
            if advance:
                self.positions[:B] += 1
                self.seq_lens[:B] += 1
            return
        self.status[2] = 3 if self.persistent_chains else 0
        self.inputs_embeds[:B] = self.text_step[:B]
        self.audio_codes[:B] = self.input_ids[:B].long()[:, None] + torch.arange(self.d.num_code_groups)[None]
        self.slot_mapping[:B] = torch.tensor([int(self.block_table[r, int(self.positions[r]) // self.block_size]) * self.block_size
                                              + int(self.positions[r]) % self.block_size for r in range(B)])
        self.last_hidden[:B] = (self.last_hidden[:B].float() + 1).to(BF16)
        # next id depends on the id, the position, the step counter and the repetition bitmap: a row restored wrongly
        # after a preemption forks its stream (the carried hidden state is left out: this fake's prefill and decode are
        # not the same function, the real engine's are)
        mix = self.input_ids[:B].long() * 7 + self.positions[:B].long() * 3 + self.steps[:B].long() * 5 \
            + self.seen[:B].sum(1).long()
        new = (mix % (self.d.codebook - 2) + 1).to(torch.int32)
        self.seen[torch.arange(B), new.long()] = 1
        self.input_ids[:B] = new
        self.steps[:B] += 1
        if advance:
            self.positions[:B] += 1
            self.seq_lens[:B] += 1
