"""TalkerEngine: device-resident weights + paged KV cache + the native decode step.

Owns what the reference's ``GPUARModelRunner`` owns for the talker stage (weights, ``kv_caches``,
persistent per-step input buffers; V/worker/gpu_ar_model_runner.py:62-72,
V/worker/gpu_model_runner.py:90-119) and drives libomni_talker.so.  Tensor-parallel layout is
Megatron-style as in vLLM: qkv / gate_up column-sharded by heads / intermediate,
o_proj / down_proj row-sharded, KV cache sharded by KV head, embed / lm_head / code predictor
replicated; one all-reduce(sum) of [B,H] bf16 after o_proj and after down_proj (RCCL).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from . import ops
from .config import TalkerDims

BF16 = torch.bfloat16


def codec_allowed_mask(d: TalkerDims, allow_eos: bool = True) -> torch.Tensor:
    """Constant logit mask (qwen3_tts_talker.py:386-394): ids [1, codebook) plus codec EOS."""
    m = torch.zeros(d.vocab, dtype=torch.uint8)
    lo, hi = 1, min(d.codebook, d.vocab)
    if hi > lo:
        m[lo:hi] = 1
    if allow_eos and 0 <= d.eos_id < d.vocab:
        m[d.eos_id] = 1
    return m


def frag_shuffle(w: torch.Tensor) -> torch.Tensor:
    """[N, K] row-major -> fragment-major (include/omni_talker.h OMNI_LAYOUT_W_FRAG): the 16-row x 32-k tile of one MFMA
    operand becomes 1 KB contiguous in lane order.  Same bytes, same shape; done once at load."""
    N, K = w.shape[-2], w.shape[-1]
    assert N % 16 == 0 and K % 32 == 0, (N, K)
    lead = w.shape[:-2]
    v = w.reshape(*lead, N // 16, 16, K // 32, 4, 8)
    nd = len(lead)
    v = v.permute(*range(nd), nd, nd + 2, nd + 3, nd + 1, nd + 4)
    return v.contiguous().reshape(*lead, N, K)


def gu8_shuffle(w: torch.Tensor) -> torch.Tensor:
    """[2I, K] = [gate rows | up rows] -> fragment-major with gate / up interleaved in groups of 8 rows
    (include/omni_talker.h OMNI_EPI_SILU_MUL_GU8): tile t = gate[8t..8t+7] then up[8t..8t+7]."""
    two_i, K = w.shape[-2], w.shape[-1]
    I = two_i // 2
    assert I % 8 == 0, I
    lead = w.shape[:-2]
    v = w.reshape(*lead, 2, I // 8, 8, K).transpose(-4, -3)        # [..., I/8, 2, 8, K]
    return frag_shuffle(v.reshape(*lead, two_i, K))


def frag_unshuffle(w: torch.Tensor) -> torch.Tensor:
    N, K = w.shape[-2], w.shape[-1]
    lead = w.shape[:-2]
    v = w.reshape(*lead, N // 16, K // 32, 4, 16, 8)
    nd = len(lead)
    v = v.permute(*range(nd), nd, nd + 3, nd + 1, nd + 2, nd + 4)
    return v.contiguous().reshape(*lead, N, K)


MOE_NAMES = ("moe_router", "moe_gate_up", "moe_down", "moe_shared_gate_up", "moe_shared_down", "moe_shared_gate")


def moe_parallel_mode(d: TalkerDims, tp: int) -> str:
    """How the routed experts are spread over a tensor-parallel group of `tp` ranks: "tp" = every expert's intermediate
    dimension split (needs whole 32-column tiles per rank), else "ep" = whole experts per rank (needs whole 16-expert groups)."""
    if tp == 1:
        return "none"
    if d.moe_shared_inter % (32 * tp):
        raise ValueError(f"shared expert width {d.moe_shared_inter} does not split into 32-column tiles over {tp} ranks")
    if d.moe_inter % (32 * tp) == 0:
        return "tp"
    if d.moe_experts % tp == 0:
        return "ep"
    raise ValueError(f"MoE of {d.moe_experts} experts x {d.moe_inter} columns does not split over {tp} ranks")


def fp8_quant_rows(w: torch.Tensor):
    """Per-output-row symmetric e4m3fn quantisation of an expert weight stack [..., rows, K]: (uint8 bytes, fp32 scales
    [..., rows]); dequantised value = bf16(fp8 * scale) (what the kernel rebuilds in registers and the oracle's weights are)."""
    wf = w.float()
    scale = (wf.abs().amax(-1).clamp_min(1e-12) / 448.0).to(torch.float32)
    q = (wf / scale[..., None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def fp8_dequant_rows(q_u8: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    return (q_u8.view(torch.float8_e4m3fn).float() * scale[..., None].float()).to(BF16)


def shard_layer(d: TalkerDims, w: dict, prefix: str, rank: int, tp: int) -> dict:
    """Per-rank slices of one backbone layer."""
    D = d.head_dim
    hq_l = d.q_heads // tp
    if tp <= d.kv_heads:
        hkv_l = d.kv_heads // tp
        kv0 = rank * hkv_l
    else:                       # more ranks than KV heads: replicate (vLLM QKVParallelLinear)
        hkv_l = 1
        kv0 = rank // (tp // d.kv_heads)
    wqkv = w[prefix + "wqkv"]
    qo, ko, vo = 0, d.q_heads * D, (d.q_heads + d.kv_heads) * D
    q = wqkv[qo + rank * hq_l * D: qo + (rank + 1) * hq_l * D]
    k = wqkv[ko + kv0 * D: ko + (kv0 + hkv_l) * D]
    v = wqkv[vo + kv0 * D: vo + (kv0 + hkv_l) * D]
    if d.moe_experts > 0:
        out = {"ln1": w[prefix + "ln1"], "ln2": w[prefix + "ln2"], "qnorm": w[prefix + "qnorm"], "knorm": w[prefix + "knorm"],
               "wqkv": torch.cat([q, k, v], 0).contiguous(),
               "wo": w[prefix + "wo"][:, rank * hq_l * D:(rank + 1) * hq_l * D].contiguous(),
               "moe_router": w[prefix + "moe_router"], "moe_shared_gate": w[prefix + "moe_shared_gate"]}
        mode, Im, Is, E = moe_parallel_mode(d, tp), d.moe_inter, d.moe_shared_inter, d.moe_experts
        gu, dn = w[prefix + "moe_gate_up"], w[prefix + "moe_down"]
        if mode == "tp":        # every expert's intermediate dimension split (vLLM FusedMoE under tensor_parallel_size)
            il = Im // tp
            out["moe_gate_up"] = torch.cat([gu[:, rank * il:(rank + 1) * il], gu[:, Im + rank * il: Im + (rank + 1) * il]], 1).contiguous()
            out["moe_down"] = dn[:, :, rank * il:(rank + 1) * il].contiguous()
        else:                   # expert parallel: a contiguous range of experts per rank (tokens are replicated on every rank)
            el = E // tp
            out["moe_gate_up"] = gu[rank * el:(rank + 1) * el].contiguous()
            out["moe_down"] = dn[rank * el:(rank + 1) * el].contiguous()
        sl = Is // tp           # the shared expert is a dense MLP: always split over its intermediate dimension
        sgu = w[prefix + "moe_shared_gate_up"]
        out["moe_shared_gate_up"] = torch.cat([sgu[rank * sl:(rank + 1) * sl], sgu[Is + rank * sl: Is + (rank + 1) * sl]], 0).contiguous()
        out["moe_shared_down"] = w[prefix + "moe_shared_down"][:, rank * sl:(rank + 1) * sl].contiguous()
        return out
    i_l = d.inter // tp
    wgu = w[prefix + "wgu"]
    return {
        "ln1": w[prefix + "ln1"], "ln2": w[prefix + "ln2"], "qnorm": w[prefix + "qnorm"], "knorm": w[prefix + "knorm"],
        "wqkv": torch.cat([q, k, v], 0).contiguous(),
        "wo": w[prefix + "wo"][:, rank * hq_l * D:(rank + 1) * hq_l * D].contiguous(),
        "wgu": torch.cat([wgu[rank * i_l:(rank + 1) * i_l], wgu[d.inter + rank * i_l: d.inter + (rank + 1) * i_l]], 0).contiguous(),
        "wdown": w[prefix + "wdown"][:, rank * i_l:(rank + 1) * i_l].contiguous(),
    }


class TalkerEngine:
    def __init__(self, dims: TalkerDims, weights: dict, *, kv_dtype: str = "fp8", num_blocks: int = 1024,
                 block_size: int = 16, max_batch: int = 64, device: str = "cuda:0", tp_rank: int = 0, tp_size: int = 1,
                 k_scale: float = 1.0, v_scale: float = 1.0, allow_eos: bool = True, masked_logit: float = 0.0, tp_group=None, n_sub: int = 1, tp_force: bool = False, frag_layout: bool = True,
                 fused_norm: bool | None = None, peer_allreduce=None, moe_fp8: bool = False, prefill_gemm: str = "tile",
                 calculate_kv_scales: bool = False):
        if not torch.cuda.is_available():
            raise L.OmniError("TalkerEngine needs an MI355X (torch.cuda unavailable); there is no CPU fallback")
        self.lib = L.load()
        self.d, self.device = dims, torch.device(device)
        self.tp_rank, self.tp_size, self.tp_group = tp_rank, tp_size, tp_group
        self.tp_path = tp_size > 1 or tp_force      # tp_force: run the collective path on a 1-rank group (tests)
        # tp_comm.PeerAllReduce (peer-mapped one-shot all-reduce fused with the residual add): the tensor-parallel rank then
        # keeps the norm-free stream and the whole step is one native call; None = RCCL all-reduces between the phase calls
        self.ar = peer_allreduce
        assert dims.q_heads % tp_size == 0 and dims.inter % tp_size == 0
        self.kv_dtype = kv_dtype
        self.frag_layout = bool(frag_layout)
        # GEMMs of the all-tokens-at-once prefill: "tile" = omni_gemm_tile on the decode step's fragment-major weights (stored
        # once); "blas" = hipBLASLt on row-major copies (another 2.8 GB at the 1.7B shape); "both" keeps both sets (tests / A-B)
        if prefill_gemm not in ("tile", "blas", "both"):
            raise ValueError(f"prefill_gemm={prefill_gemm!r}")
        self.prefill_gemm = prefill_gemm if self.frag_layout else "blas"
        # norm-free residual stream (omni_gemm_resid / omni_gemm_xnorm; MoE layers: omni_moe_experts_resid): single rank, or
        # tensor- / expert-parallel ranks with the peer-mapped all-reduce (it sums the partials BEFORE the residual add);
        # ranks that all-reduce through RCCL between the phase calls keep the separate norms
        self.fused_norm = (self.frag_layout and (not self.tp_path or self.ar is not None)) \
            if fused_norm is None else bool(fused_norm)
        if dims.moe_experts > 0 and not self.frag_layout:
            raise ValueError("the sparse-MoE backbone needs the fragment-major layout")
        self.moe_fp8 = bool(moe_fp8) and dims.moe_experts > 0
        self.moe_mode = moe_parallel_mode(dims, tp_size) if dims.moe_experts > 0 else "none"
        if self.fused_norm and ((self.tp_path and self.ar is None) or not self.frag_layout):
            raise ValueError("fused_norm needs frag_layout and a single rank (or the peer-mapped all-reduce)")
        if self.ar is not None and (not self.fused_norm or self.ar.world != tp_size or self.ar.rank != tp_rank or self.ar.hidden != dims.hidden
                                    or self.ar.rows16 < (max_batch + 15) // 16 * 16):
            raise ValueError("peer_allreduce does not match this engine (world / rank / hidden / rows, fused_norm)")
        self.kv_code = L.KV_CODES[kv_dtype]
        self.block_size, self.num_blocks, self.max_batch = block_size, num_blocks, max_batch
        self.hq_l = dims.q_heads // tp_size
        self.hkv_l = max(dims.kv_heads // tp_size, 1)
        self.inter_l = dims.inter // tp_size
        self.moe_inter_l = dims.moe_inter // tp_size if dims.moe_experts > 0 and self.moe_mode == "tp" else dims.moe_inter
        self.moe_experts_l = dims.moe_experts // tp_size if dims.moe_experts > 0 and self.moe_mode == "ep" else dims.moe_experts
        self.moe_e0 = tp_rank * self.moe_experts_l if self.moe_mode == "ep" else 0
        self.moe_shared_l = dims.moe_shared_inter // tp_size
        self.bt_stride = (dims.max_model_len + block_size - 1) // block_size
        dev = self.device
        d = dims

        # ---- weights -> HBM
        self._keep: list[torch.Tensor] = []

        def up(t: torch.Tensor) -> torch.Tensor:
            t = t.to(dev).contiguous()
            self._keep.append(t)
            return t

        self.embed = up(weights["embed"])
        self.final_norm = up(weights["norm"])
        self.lm_head = up(weights["lm_head"])
        self.allowed = up(codec_allowed_mask(d, allow_eos))
        # rows of the rotary table: max_model_len, and as many again for M-RoPE models (a prompt's ids may run ahead of its token
        # count and every later position is index + mrope_position_delta; the runner refuses requests that would leave the table)
        self.rope_rows = d.max_model_len * (2 if d.mrope_section else 1)
        self.cos_sin = up(ops.rope_table(self.rope_rows, d.head_dim, d.rope_theta))
        self.cp_cos_sin = up(ops.rope_table(d.num_code_groups + 1, d.cp_head_dim, d.cp_rope_theta))
        self.cp_norm = up(weights["cp.norm"])
        self.cp_lm_head = up(weights["cp.lm_head"])
        self.cp_embed = up(weights["cp.embed"])
        self.cp_proj_w = up(weights["cp.proj_w"]) if d.has_cp_projection else None
        self.cp_proj_b = up(weights["cp.proj_b"]) if d.has_cp_projection else None
        # constant-folded projections of the embedding tables (bit-identical to projecting at run time:
        # the GEMM result of a row does not depend on which other rows share the launch)
        if d.has_cp_projection:
            def fold(table: torch.Tensor) -> torch.Tensor:
                out = torch.empty(table.shape[0], d.cp_hidden, dtype=BF16, device=dev)
                for r0 in range(0, table.shape[0], 64):
                    out[r0:r0 + 64] = ops.gemm(table[r0:r0 + 64].contiguous(), self.cp_proj_w, bias=self.cp_proj_b)
                return out
            self.cp_proj_table = torch.stack([fold(self.cp_embed[g]) for g in range(d.num_code_groups - 1)]).contiguous()
            self.cp_e0_table = fold(self.embed)
            torch.cuda.synchronize()
        else:
            self.cp_proj_table, self.cp_e0_table = self.cp_embed, self.embed
        names = ("ln1", "wqkv", "qnorm", "knorm", "wo", "ln2", "wgu", "wdown")
        bb_names = names[:6] + MOE_NAMES if d.moe_experts > 0 else names
        frag_names = ("wqkv", "wo", "wgu", "wdown", "moe_gate_up", "moe_down", "moe_shared_gate_up", "moe_shared_down")
        if self.fused_norm:
            frag_names += ("moe_router",)      # takes the fused-norm prologue on the norm-free stream (omni_gemm_xnorm: W fragment-major)
        self._layers = (L.LayerWeights * d.layers)()
        self._frag_keep: dict = {}
        self.layer_w: list[dict] = []
        for i in range(d.layers):
            sh = shard_layer(d, weights, f"l{i}.", tp_rank, tp_size)
            if self.moe_fp8:
                # fp8 e4m3fn expert weights (BASELINE config #5): quantised per output row at load; everything downstream
                # -- the decode kernels' in-register dequantisation, the prefill's bf16 copies, the oracle in the tests --
                # sees the SAME matrix bf16(fp8 * scale)
                for n in ("moe_gate_up", "moe_down"):
                    q8, sc = fp8_quant_rows(sh[n])
                    sh[n + "_q8"], sh[n + "_scale"] = q8, sc
                    sh[n] = fp8_dequant_rows(q8, sc)
            # row-major copies: norms always; the dense GEMM weights only for the hipBLASLt prefill; MoE tensors for its batched path
            dense = ("wqkv", "wo", "wgu", "wdown")
            lw = {n: up(sh[n]) for n in bb_names if self.prefill_gemm != "tile" or n not in dense}
            self.layer_w.append(lw)
            for n in bb_names:
                if self.frag_layout and n in frag_names:
                    # fragment-major copy for the native GEMMs (per expert for MoE; dense gate_up interleaved by 8)
                    if self.moe_fp8 and n in ("moe_gate_up", "moe_down"):
                        t_ = up(frag_shuffle(sh[n + "_q8"]))            # same element order, one byte per weight
                        setattr(self._layers[i], n + "_scale", up(sh[n + "_scale"]).data_ptr())
                    else:
                        src = lw[n] if n in lw else sh[n].to(dev)       # shuffled on the device; the row-major temporary is dropped
                        t_ = up(gu8_shuffle(src) if n == "wgu" else frag_shuffle(src))
                        del src
                    if n in dense:
                        lw[n + "_f"] = t_
                    elif not (self.moe_fp8 and n in ("moe_gate_up", "moe_down")):
                        self._frag_keep[(i, n)] = t_
                else:
                    t_ = lw[n]
                setattr(self._layers[i], n, t_.data_ptr())
            if d.moe_experts > 0 and self.prefill_gemm in ("tile", "both"):
                # the prefill's grouped omni_gemm_tile reads the decode step's fragment-major expert matrices (bf16 experts: the
                # very same tensors; fp8 experts: one dequantised fragment-major copy -- the tile kernel's LDS-DMA path moves
                # bf16 fragments); router / shared-expert gate as fragment-major too (the gate's single row padded to one tile)
                for n in ("moe_gate_up", "moe_down"):
                    lw[n + "_f"] = up(frag_shuffle(lw[n])) if self.moe_fp8 else self._frag_keep[(i, n)]
                lw["moe_router_f"] = up(frag_shuffle(lw["moe_router"]))
                if self.moe_shared_l > 0:
                    for n in ("moe_shared_gate_up", "moe_shared_down"):
                        lw[n + "_f"] = self._frag_keep[(i, n)]
                    g16 = torch.zeros(16, d.hidden, dtype=BF16, device=dev)
                    g16[:1] = lw["moe_shared_gate"].reshape(1, -1)
                    lw["moe_shared_gate_f"] = frag_shuffle(g16)
            if d.moe_experts > 0:
                # the batched hipBLASLt prefill (prefill_gemm="blas", kept for A/B) wants [E, K, N] operands stored that way:
                # torch.bmm on the transposed VIEW of [E, N, K] faults on this ROCm build at the talker's shapes
                # (scripts/diag_bmm.py); they replace the row-major expert copies, which nothing else reads
                for n in ("moe_gate_up", "moe_down"):
                    rm = lw.pop(n)
                    if self.prefill_gemm in ("blas", "both"):
                        lw[n + "_t"] = rm.transpose(1, 2).contiguous()
                    del rm
        self._cp_layers = (L.LayerWeights * d.cp_layers)()
        self.cp_layer_w: list[dict] = []
        for i in range(d.cp_layers):
            lw = {n: up(weights[f"cp.l{i}.{n}"]) for n in names}
            if self.frag_layout:
                for n in ("wqkv", "wo", "wgu", "wdown"):
                    lw[n] = up(gu8_shuffle(lw[n]) if n == "wgu" else frag_shuffle(lw[n]))
            self.cp_layer_w.append(lw)
            for n in names:
                setattr(self._cp_layers[i], n, lw[n].data_ptr())

        # ---- paged KV cache, one tensor per layer: [2, num_blocks, block_size, Hkv_local, D]
        self._alloc_kv(num_blocks)

        # ---- descriptor + scratch + native engine
        desc = L.TalkerDesc()
        desc.hidden, desc.layers, desc.q_heads, desc.kv_heads = d.hidden, d.layers, self.hq_l, self.hkv_l
        desc.head_dim, desc.inter, desc.vocab, desc.codebook = d.head_dim, self.inter_l, d.vocab, d.codebook
        desc.num_code_groups, desc.eps = d.num_code_groups, d.eps
        desc.cp_hidden, desc.cp_layers, desc.cp_q_heads, desc.cp_kv_heads = d.cp_hidden, d.cp_layers, d.cp_q_heads, d.cp_kv_heads
        desc.cp_head_dim, desc.cp_inter, desc.has_cp_projection = d.cp_head_dim, d.cp_inter, int(d.has_cp_projection)
        desc.frag_layout = int(self.frag_layout)
        desc.fused_norm = int(self.fused_norm)
        # the code predictor is replicated on every rank and has no collective inside: norm-free whenever layouts allow
        self.cp_fused_norm = self.frag_layout if fused_norm is None else bool(fused_norm)
        desc.cp_fused_norm = int(self.cp_fused_norm)
        desc.moe_experts, desc.moe_top_k, desc.moe_inter = d.moe_experts, d.moe_top_k, self.moe_inter_l
        desc.moe_shared_inter, desc.moe_norm_topk = self.moe_shared_l, int(d.moe_norm_topk)
        desc.moe_e0, desc.moe_experts_local, desc.moe_w8 = self.moe_e0, self.moe_experts_l if d.moe_experts > 0 else 0, int(self.moe_fp8)
        # the persistent code-predictor chain needs its 256-workgroup grid co-resident: not for engines whose steps run
        # concurrently on one GPU (n_sub parallel graph branches)
        desc.cp_chain = int(int(n_sub) <= 1 and os.environ.get("OMNI_CP_CHAIN", "1") != "0")
        desc.rope_rows = self.rope_rows
        self.persistent_chains = bool(desc.cp_chain)
        if self.frag_layout:        # GEMM weights the native step reads: fragment-major device copies
            self._lm_head_f = up(frag_shuffle(self.lm_head))
            self._cp_lm_head_f = up(frag_shuffle(self.cp_lm_head))
            self._cp_proj_w_f = up(frag_shuffle(self.cp_proj_w)) if self.cp_proj_w is not None else None
        # more than 64 rows only as n_sub concurrent row ranges (each <= 64: the skinny GEMM's M limit)
        desc.max_batch, desc.block_size, desc.kv_dtype = (min(max_batch, 64) if int(n_sub) > 1 else max_batch), block_size, self.kv_code
        desc.max_model_len, desc.bt_stride = d.max_model_len, self.bt_stride
        desc.k_scale, desc.v_scale = k_scale, v_scale
        # fp8 KV scales per layer (host copy = what the eager prefill launches pass; the native engine keeps a device table for the
        # captured decode steps).  calculate_kv_scales (vLLM cache_config; the reference's first forward runs eager for it,
        # V/worker/gpu_ar_model_runner.py:122,269-275): the FIRST prefill pass of this engine sets k = max|k| / 200, v = max|v| / 100 per
        # layer from its own tokens, before its cache write (oracle: talker_oracle.TalkerOracle(calculate_kv_scales=True))
        self.k_scale_l, self.v_scale_l = [float(k_scale)] * d.layers, [float(v_scale)] * d.layers
        self._calibrate_requested = bool(calculate_kv_scales)
        self.calibrate_pending = bool(calculate_kv_scales) and self.kv_code == L.KV_FP8
        desc.masked_logit = float(masked_logit)      # 0: -inf (TTS); -1e9: the Omni talker's finite suppression value
        desc.embed, desc.final_norm = self.embed.data_ptr(), self.final_norm.data_ptr()
        desc.layer = C.cast(self._layers, C.POINTER(L.LayerWeights))
        desc.lm_head = (self._lm_head_f if self.frag_layout else self.lm_head).data_ptr()
        desc.allowed_mask, desc.cos_sin = self.allowed.data_ptr(), self.cos_sin.data_ptr()
        desc.cp_proj_w = L.ptr(self._cp_proj_w_f if self.frag_layout else self.cp_proj_w)
        desc.cp_proj_b = L.ptr(self.cp_proj_b)
        desc.cp_layer = C.cast(self._cp_layers, C.POINTER(L.LayerWeights))
        desc.cp_norm = self.cp_norm.data_ptr()
        desc.cp_lm_head = (self._cp_lm_head_f if self.frag_layout else self.cp_lm_head).data_ptr()
        desc.cp_embed, desc.cp_cos_sin = self.cp_embed.data_ptr(), self.cp_cos_sin.data_ptr()
        desc.cp_proj_table, desc.cp_e0_table = self.cp_proj_table.data_ptr(), self.cp_e0_table.data_ptr()
        pvp = C.POINTER(C.c_void_p)
        desc.k_cache, desc.v_cache = C.cast(self._kc, pvp), C.cast(self._vc, pvp)
        if self._ks is not None:
            desc.k_scales, desc.v_scales = C.cast(self._ks, pvp), C.cast(self._vs, pvp)
        if self.ar is not None:
            desc.ar_attn, desc.ar_mlp = C.pointer(self.ar.peers[0]), C.pointer(self.ar.peers[1])
        nbytes = self.lib.omni_talker_scratch_bytes(C.byref(desc))
        if nbytes < 0:
            raise L.OmniError("omni_talker_scratch_bytes: " + self.lib.omni_last_error().decode())
        self.scratch = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
        desc.scratch, desc.scratch_bytes = self.scratch.data_ptr(), nbytes
        self._desc = desc
        self.handle = self.lib.omni_talker_create(C.byref(desc))
        if not self.handle:
            raise L.OmniError("omni_talker_create: " + self.lib.omni_last_error().decode())
        # ---- sub-batch engines: the decode step is latency-bound (dependent small launches), so independent row ranges
        # of the batch run as concurrent branches of the same hipGraph on separate HIP streams; they share weights and
        # the KV pool and differ only in their scratch (activations, code-predictor KV) and row offset.
        self.n_sub = max(1, int(n_sub)) if not self.tp_path else 1
        self.sub_rows = (max_batch + self.n_sub - 1) // self.n_sub
        self._sub = []
        if self.n_sub > 1:
            for k in range(self.n_sub):
                dk = L.TalkerDesc.from_buffer_copy(desc)
                dk.max_batch = self.sub_rows
                nb2 = self.lib.omni_talker_scratch_bytes(C.byref(dk))
                sc = torch.zeros(nb2, dtype=torch.uint8, device=dev)
                dk.scratch, dk.scratch_bytes = sc.data_ptr(), nb2
                h = self.lib.omni_talker_create(C.byref(dk))
                if not h:
                    raise L.OmniError("omni_talker_create(sub): " + self.lib.omni_last_error().decode())
                self._sub.append((h, sc, dk))
            self._sub_streams = [torch.cuda.Stream(device=dev) for _ in range(self.n_sub)]
            # (round 6: TWO concurrent row ranges CAN keep their backbones' persistent launches on half grids -- set_chains(2), 128 workgroups
            #  each, co-resident, same bits: tests/test_gpu_tp_chain.py -- but measured slower than launch per op at 2 x 64 rows, 24.3 k against
            #  25.3 k tokens/s: half the chip per stage and the other range's kernels beside it; so the ranges stay launch per op by default)

        # ---- persistent per-step buffers (graph-stable addresses), row r = batch slot r
        Bm, H, Q = max_batch, d.hidden, d.num_code_groups
        z = lambda *s, dt: torch.zeros(*s, dtype=dt, device=dev)  # noqa: E731
        # input_ids [Bm] and the step's four status words (omni_step_io.status: chain error word, peer all-reduce error word,
        # which chains ran, 0) share ONE tensor: the runner's per-step host copy of the sampled ids brings the status along
        # ... and ids + status, the step's code frames and h[t] share ONE allocation, the step's OUTPUT RECORD: what the runner
        # hands on per step is one copy (async scheduling: one device-to-device snapshot behind the step, runner._snapshot)
        al = lambda n: (n + 255) // 256 * 256  # noqa: E731
        self._rec_off = (0, al((Bm + 4) * 4), al((Bm + 4) * 4) + al(Bm * Q * 8))
        self.out_record = z(self._rec_off[2] + Bm * H * 2, dt=torch.uint8)
        self._op_handle = None                  # torch_ops.register_engine(self), on first use
        self.ids_status, self.audio_codes, self.last_hidden = self.split_out_record(self.out_record)
        self.input_ids = self.ids_status[:Bm]
        self.status = self.ids_status[Bm:]
        self.positions = z(Bm, dt=torch.int32)
        self.seq_lens = z(Bm, dt=torch.int32)
        self.block_table = z(Bm, self.bt_stride, dt=torch.int32)
        self.slot_mapping = z(Bm, dt=torch.int64)
        self.text_step = z(Bm, H, dt=BF16)
        self.inputs_embeds = z(Bm, H, dt=BF16)
        self.logits = z(Bm, d.vocab, dt=torch.float32)
        self.seen = z(Bm, d.vocab, dt=torch.uint8)
        self.steps = z(Bm, dt=torch.int32)
        # per-request sampling parameters as per-row device arrays read inside the captured sampler launch
        # (V/worker/gpu_model_runner.py:315-319: one SamplingParams + generator per request): a graph serves any mix
        self.row_greedy = torch.ones(Bm, dtype=torch.int32, device=dev)
        self.row_temperature = torch.ones(Bm, dtype=torch.float32, device=dev)
        self.row_top_k = z(Bm, dt=torch.int32)
        self.row_top_p = torch.ones(Bm, dtype=torch.float32, device=dev)
        self.row_rep_penalty = torch.ones(Bm, dtype=torch.float32, device=dev)
        self.row_seed = z(Bm, dt=torch.int32)               # uint32 bit patterns
        # live decode rows of the step: rows [num_live, B) of a padded graph bucket are inert (no KV write / sample / advance)
        self.num_live = torch.full((1,), Bm, dtype=torch.int32, device=dev)
        # M-RoPE models: the axis table of the prefill kernel and every row's rotary offset for the decode steps (a request's
        # mrope_position_delta; 0 for the talker's usual three identical id rows)
        self.mrope_axis = ops.mrope_axis_table(d.mrope_section, d.mrope_interleaved).to(dev) if d.mrope_section else None
        self.rope_delta = torch.zeros(Bm, dtype=torch.int32, device=dev) if d.mrope_section else None
        # launch-wide scalars: the code predictor's model-level parameters (qwen3_tts_talker.py:1620-1627); the layer-0
        # entries mirror what set_sampling last broadcast into the row arrays
        self.sampling = dict(greedy=1, temperature=1.0, top_k=0, top_p=1.0, rep_penalty=1.0, seed=0, cp_greedy=1,
                             cp_temperature=0.9, cp_top_k=50, cp_top_p=1.0)
        self._attn_out = self._scratch_view(self.lib.omni_talker_attn_out(self.handle), Bm * H).view(Bm, H)
        self._mlp_out = self._scratch_view(self.lib.omni_talker_mlp_out(self.handle), Bm * H).view(Bm, H)

    def _alloc_kv(self, num_blocks: int) -> None:
        d, dev = self.d, self.device
        store = {"bf16": BF16, "auto": BF16, "fp8": torch.uint8, "fp8_e4m3": torch.uint8, "int8": torch.int8, "fp16": torch.float16,
                 "float16": torch.float16, "half": torch.float16}[self.kv_dtype]
        shape = (2, num_blocks, self.block_size, self.hkv_l, d.head_dim)
        self.num_blocks = num_blocks
        self.kv_caches = [torch.zeros(shape, dtype=store, device=dev) for _ in range(d.layers)]
        self.kv_scales = ([torch.zeros(shape[:-1], dtype=torch.float32, device=dev) for _ in range(d.layers)]
                          if self.kv_code == L.KV_INT8 else None)
        self._kc = (C.c_void_p * d.layers)(*[c[0].data_ptr() for c in self.kv_caches])
        self._vc = (C.c_void_p * d.layers)(*[c[1].data_ptr() for c in self.kv_caches])
        if self.kv_scales is not None:
            self._ks = (C.c_void_p * d.layers)(*[s[0].data_ptr() for s in self.kv_scales])
            self._vs = (C.c_void_p * d.layers)(*[s[1].data_ptr() for s in self.kv_scales])
        else:
            self._ks = self._vs = None

    def kv_cache_bytes(self) -> int:
        return sum(c.numel() * c.element_size() for c in self.kv_caches) + sum(s.numel() * 4 for s in (self.kv_scales or []))

    def resize_kv_cache(self, num_blocks: int) -> None:
        """Replace the paged KV cache by one of `num_blocks` blocks, keeping weights, scratch and every per-step buffer: the worker
        measures the process's footprint on a probe cache first (worker.determine_available_memory), then sizes the real one.  The native
        engine is re-created on the same descriptor (its cache pointers are creation-time constants); call before any graph capture."""
        if self.n_sub > 1:
            raise L.OmniError("resize_kv_cache: single-handle engines only")
        self.lib.omni_talker_destroy(self.handle)
        self.handle = None
        self.kv_caches = self.kv_scales = None
        torch.cuda.empty_cache()
        self._alloc_kv(int(num_blocks))
        pvp = C.POINTER(C.c_void_p)
        self._desc.k_cache, self._desc.v_cache = C.cast(self._kc, pvp), C.cast(self._vc, pvp)
        if self._ks is not None:
            self._desc.k_scales, self._desc.v_scales = C.cast(self._ks, pvp), C.cast(self._vs, pvp)
        self.scratch.zero_()
        self.handle = self.lib.omni_talker_create(C.byref(self._desc))
        if not self.handle:
            raise L.OmniError("omni_talker_create: " + self.lib.omni_last_error().decode())
        self._op_handle = None
        self.calibrate_pending = bool(getattr(self, "_calibrate_requested", False)) and self.kv_code == L.KV_FP8

    def profile_run(self, num_tokens: int, batch: int) -> None:
        """The worker's memory probe (V/worker/base.py:118-123 `profile_run`): one prefill of `num_tokens` synthetic prompt tokens spread
        over `batch` requests and one decode step at `batch` rows, so that every workspace, code object and allocator pool the steps
        need exists when the footprint is read.  The probe cache must hold the tokens; its contents are garbage afterwards."""
        d, bs = self.d, self.block_size
        batch = max(1, min(batch, self.max_batch))
        per = max(1, min(num_tokens // batch, d.max_model_len - 2, (self.num_blocks - 1) // batch * bs - 1))
        lens = [per] * batch
        bt = torch.zeros(self.max_batch, self.bt_stride, dtype=torch.int32)
        nxt = 1
        for r, n in enumerate(lens):
            need = (n + 1 + bs - 1) // bs
            bt[r, :need] = torch.arange(nxt, nxt + need, dtype=torch.int32)
            nxt += need
        self.block_table.copy_(bt)
        T = sum(lens)
        x = torch.zeros(T, d.hidden, dtype=BF16, device=self.device)
        pos = torch.cat([torch.arange(n) for n in lens]).to(torch.int32)
        req = torch.cat([torch.full((n,), r) for r, n in enumerate(lens)]).to(torch.int32)
        slots = torch.tensor([int(bt[int(req[t]), int(pos[t]) // bs]) * bs + int(pos[t]) % bs for t in range(T)])
        cal, self.calibrate_pending = self.calibrate_pending, False          # the probe must not fix the fp8 scales
        self.prefill(x, pos.to(self.device), req.to(self.device), slots.to(self.device))
        self.positions[:batch] = per
        self.seq_lens[:batch] = per + 1
        self.decode_step(batch)
        torch.cuda.synchronize(self.device)
        self.calibrate_pending = cal
        for t in (self.positions, self.seq_lens, self.steps, self.seen, self.block_table, self.input_ids):
            t.zero_()

    def split_out_record(self, rec: torch.Tensor):
        """(ids + 4 status words [Bm + 4] int32, audio codes [Bm, Q] int64, h [Bm, H] bf16) as views of an output record --
        the engine's own (device) or a host copy of it."""
        Bm, H, Q = self.max_batch, self.d.hidden, self.d.num_code_groups
        o0, o1, o2 = self._rec_off
        return (rec[o0:o0 + (Bm + 4) * 4].view(torch.int32), rec[o1:o1 + Bm * Q * 8].view(torch.int64).view(Bm, Q),
                rec[o2:o2 + Bm * H * 2].view(BF16).view(Bm, H))

    def _scratch_view(self, p: int, n_bf16: int) -> torch.Tensor:
        off = p - self.scratch.data_ptr()
        return self.scratch[off: off + 2 * n_bf16].view(BF16)

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            self.lib.omni_talker_destroy(h)
        for hk, _, _ in getattr(self, "_sub", []):
            self.lib.omni_talker_destroy(hk)
        self._sub = []

    # ------------------------------------------------------------------ step
    ROW_KEYS = ("greedy", "temperature", "top_k", "top_p", "rep_penalty", "seed")

    def set_sampling(self, **kw) -> None:
        """Layer-0 parameters are broadcast to every row's device entry; cp_* are the launch-wide code-predictor scalars
        (baked into a captured graph: change them before capture)."""
        self.sampling.update(kw)
        for k in self.ROW_KEYS:
            if k in kw:
                self._row_fill(k, slice(None), kw[k])

    def _row_fill(self, key: str, rows, value) -> None:
        buf = getattr(self, "row_" + key)
        if key == "seed":
            value = int(value) & 0xFFFFFFFF
            value = value - (1 << 32) if value >= (1 << 31) else value      # uint32 bit pattern in an int32 tensor
        elif key == "greedy":
            value = int(bool(value))
        buf[rows] = value

    def set_row_sampling(self, row: int, *, greedy, temperature, top_k, top_p, rep_penalty, seed) -> None:
        """One request's SamplingParams into its batch row (runner: on admission / row moves ride with _ROW_BUFFERS)."""
        if not greedy and not temperature > 0.0:
            raise ValueError("temperature must be > 0 when sampling")
        if 0.0 < top_p < 1.0 and not 0 < top_k <= 1024:
            raise ValueError("top_p < 1 needs 0 < top_k <= 1024 on this path")
        if not rep_penalty > 0.0:
            raise ValueError("repetition_penalty must be > 0")
        for k, v in (("greedy", greedy), ("temperature", temperature if temperature > 0 else 1.0), ("top_k", top_k), ("top_p", top_p),
                     ("rep_penalty", rep_penalty), ("seed", seed)):
            self._row_fill(k, row, v)

    def _io(self, B: int, advance: bool, row0: int = 0) -> L.StepIO:
        io = L.StepIO()
        io.B = B
        for n in ("input_ids", "positions", "seq_lens", "block_table", "slot_mapping", "last_hidden", "text_step",
                  "inputs_embeds", "audio_codes", "logits", "steps"):
            setattr(io, n, getattr(self, n)[row0:].data_ptr())
        s = self.sampling
        io.seen = self.seen[row0:].data_ptr()       # always: a row's repetition penalty may differ from its neighbours'
        io.greedy, io.temperature, io.top_k = int(s["greedy"]), float(s["temperature"]), int(s["top_k"])
        io.rep_penalty, io.seed = float(s["rep_penalty"]), int(s["seed"]) & 0xFFFFFFFF
        io.cp_greedy, io.cp_temperature, io.cp_top_k = int(s["cp_greedy"]), float(s["cp_temperature"]), int(s["cp_top_k"])
        io.advance = int(advance)
        io.top_p, io.cp_top_p = float(s.get("top_p", 1.0)), float(s.get("cp_top_p", 1.0))
        for k in self.ROW_KEYS:
            setattr(io.rows, k, getattr(self, "row_" + k)[row0:].data_ptr())
        # sub-batch branches see their own row range: their live count is the bucket (only the runner path pads buckets)
        io.num_live = self.num_live.data_ptr() if row0 == 0 and self.n_sub == 1 else None
        io.rope_delta = self.rope_delta[row0:].data_ptr() if self.rope_delta is not None else None
        io.status = self.status.data_ptr() if row0 == 0 and self.n_sub == 1 else None
        return io

    def decode_step(self, B: int, advance: bool = True) -> None:
        """One talker decode step for rows [0, B) on torch's current stream (capturable): `torch.ops.mi355x_omni.decode_step_`
        over omni_talker_decode_step -- host code calling HIP through a PyTorch-ROCm custom op, for the one call that matters."""
        if self._op_handle is None:
            from . import torch_ops
            self._op_handle = torch_ops.register_engine(self)
        torch.ops.mi355x_omni.decode_step_(self._op_handle, self.out_record, self.positions, self.seq_lens, self.seen, self.steps,
                                           self.kv_caches, self.text_step, self.block_table, self.num_live, int(B), bool(advance))

    def _decode_step_native(self, B: int, advance: bool = True) -> None:
        st = L.current_stream()
        if self.n_sub > 1 and B > self.sub_rows:
            # fork: one branch per row range (works eagerly and under stream capture -> parallel graph branches)
            cur = torch.cuda.current_stream()
            fork = torch.cuda.Event()
            fork.record(cur)
            joins = []
            for k in range(self.n_sub):
                r0 = k * self.sub_rows
                rows = min(self.sub_rows, B - r0)
                if rows <= 0:
                    break
                sk = self._sub_streams[k]
                sk.wait_event(fork)
                io = self._io(rows, advance, r0)
                L.check(self.lib.omni_talker_decode_step(self._sub[k][0], C.byref(io), sk.cuda_stream), "omni_talker_decode_step(sub)")
                ev = torch.cuda.Event()
                ev.record(sk)
                joins.append(ev)
            for ev in joins:
                cur.wait_event(ev)
            return
        io = self._io(B, advance)
        if not self.tp_path or self.ar is not None:
            L.check(self.lib.omni_talker_decode_step(self.handle, C.byref(io), st), "omni_talker_decode_step")
            return
        import torch.distributed as dist
        L.check(self.lib.omni_talker_mtp(self.handle, C.byref(io), st), "omni_talker_mtp")
        for l in range(self.d.layers):
            L.check(self.lib.omni_talker_layer_attn(self.handle, C.byref(io), l, st), "omni_talker_layer_attn")
            dist.all_reduce(self._attn_out[:B], group=self.tp_group)
            L.check(self.lib.omni_talker_layer_mlp(self.handle, C.byref(io), l, st), "omni_talker_layer_mlp")
            dist.all_reduce(self._mlp_out[:B], group=self.tp_group)
        L.check(self.lib.omni_talker_finish(self.handle, C.byref(io), st), "omni_talker_finish")

    def backbone_step(self, B: int) -> None:
        """Diagnostics: the backbone half of a decode step alone (28 layers + lm_head + sampler, no code predictor /
        input assembly, positions not advanced) on whatever residual stream the last step left -- timing only."""
        if self.tp_path:
            raise L.OmniError("backbone_step: single-rank diagnostic")
        io = self._io(B, False)
        L.check(self.lib.omni_talker_backbone_step(self.handle, C.byref(io), L.current_stream()), "omni_talker_backbone_step")

    def step_part(self, B: int, parts: int) -> None:
        """Timing attribution: launch only `parts` of a decode step (1 mtp phase, 2 attention launches, 4 rest of the backbone,
        8 lm_head + sampler); positions are not advanced.  Outputs of a partial step are meaningless."""
        if self.tp_path:
            raise L.OmniError("step_part: single-rank diagnostic")
        io = self._io(B, False)
        L.check(self.lib.omni_talker_step_part(self.handle, C.byref(io), int(parts), L.current_stream()), "omni_talker_step_part")

    def prefill_wide(self, x: torch.Tensor, positions: torch.Tensor, req_of_tok: torch.Tensor, slot_mapping: torch.Tensor,
                     block_table: torch.Tensor | None = None, gemm: str | None = None, rope_positions: torch.Tensor | None = None) -> torch.Tensor:
        """Prefill of all T prompt tokens in one pass per layer (instead of max_batch-row chunks that each re-stream the
        weights): norm, q/k-norm + RoPE + KV write, causal paged attention on the native kernels; the four per-layer GEMMs
        either on omni_gemm_tile (gemm = "tile": the decode step's fragment-major weights, SiLU(gate) * up fused into the
        gate_up epilogue) or on hipBLASLt (gemm = "blas": torch.nn.functional.linear on row-major copies).  Both: bf16 in,
        fp32 accumulate, one rounding -- the skinny kernel's convention."""
        import torch.nn.functional as F
        d = self.d
        gemm = ("tile" if self.prefill_gemm in ("tile", "both") else "blas") if gemm is None else gemm
        if (gemm == "tile" and self.prefill_gemm == "blas") or (gemm == "blas" and self.prefill_gemm == "tile"):
            raise L.OmniError(f"this engine was built with prefill_gemm={self.prefill_gemm!r}: no weights for the {gemm} prefill")
        tile = gemm == "tile"
        bt = self.block_table if block_table is None else block_table
        D, hq, hkv = d.head_dim, self.hq_l, self.hkv_l
        resid = x.clone()
        delta = None
        # rope_positions int32 [3, T]: M-RoPE ids (positions.get_input_positions_tensor) that may differ between rows and from the
        # cache positions; `positions` stays the token's index in its sequence (cache slot order, causal mask)
        rope_pos, axis = positions, None
        if rope_positions is not None:
            if self.mrope_axis is None:
                raise L.OmniError("rope_positions given but the model has no mrope_section")
            # (ids outside the rotary table are refused at admission, runner._update_states; the clamp is the backstop)
            rope_pos, axis = rope_positions.to(torch.int32).clamp(0, self.rope_rows - 1).contiguous(), self.mrope_axis
        calibrate = self.calibrate_pending
        if calibrate:
            T = x.shape[0]
            tmp_k = torch.empty(T, hkv, D, dtype=BF16, device=x.device)      # this pass's K (after q/k-norm + RoPE) and V, unquantised
            tmp_v = torch.empty_like(tmp_k)
            tmp_slots = torch.arange(T, dtype=torch.int64, device=x.device)
        for l in range(d.layers):
            w = self.layer_w[l]
            a = ops.rmsnorm(None, w["ln1"], d.eps, delta=delta, residual=resid)
            qkv = ops.gemm_tile(a, w["wqkv_f"]) if tile else F.linear(a, w["wqkv"])
            kc, vc = self.kv_caches[l][0], self.kv_caches[l][1]
            ks = self.kv_scales[l] if self.kv_scales is not None else None
            if calibrate:
                # the same kernel into a bf16 scratch "cache" (slot t = token t): max|k|, max|v| of this pass -> this layer's scales
                ops.qknorm_rope_kvwrite(qkv, w["qnorm"], w["knorm"], rope_pos, self.cos_sin, tmp_slots, tmp_k, tmp_v, q_heads=hq,
                                        kv_heads=hkv, head_dim=D, eps=d.eps, kv_dtype=L.KV_BF16, mrope_axis=axis)
                ka, va = tmp_k.float().abs().max(), tmp_v.float().abs().max()
                self.k_scale_l[l] = float(ka / 200.0) if float(ka) > 0 else 1.0          # fp32 divisions (the oracle's)
                self.v_scale_l[l] = float(va / 100.0) if float(va) > 0 else 1.0
            q = ops.qknorm_rope_kvwrite(qkv, w["qnorm"], w["knorm"], rope_pos, self.cos_sin, slot_mapping, kc, vc,
                                        q_heads=hq, kv_heads=hkv, head_dim=D, eps=d.eps, kv_dtype=self.kv_code,
                                        k_scale=self.k_scale_l[l], v_scale=self.v_scale_l[l],
                                        k_scales=None if ks is None else ks[0], v_scales=None if ks is None else ks[1],
                                        mrope_axis=axis)
            o = ops.paged_attn_prefill(q, kc, vc, bt, req_of_tok, positions, q_heads=hq, kv_heads=hkv, head_dim=D,
                                       block_size=self.block_size, kv_dtype=self.kv_code, k_scale=self.k_scale_l[l],
                                       v_scale=self.v_scale_l[l], k_scales=None if ks is None else ks[0],
                                       v_scales=None if ks is None else ks[1])
            o = ops.gemm_tile(o, w["wo_f"]) if tile else F.linear(o, w["wo"])
            if self.tp_path:
                torch.distributed.all_reduce(o, group=self.tp_group)
            a = ops.rmsnorm(None, w["ln2"], d.eps, delta=o, residual=resid)
            if d.moe_experts > 0:
                delta = self._moe_mlp_tile(a, w) if tile else self._moe_mlp_blas(a, w)
            elif tile:
                delta = ops.gemm_tile(ops.gemm_tile(a, w["wgu_f"], act=L.TILE_ACT_SILU_MUL_GU8), w["wdown_f"])
            else:
                delta = F.linear(ops.silu_mul(F.linear(a, w["wgu"])), w["wdown"])
            if self.tp_path:
                torch.distributed.all_reduce(delta, group=self.tp_group)
        if calibrate:
            self.set_kv_scales(self.k_scale_l, self.v_scale_l)
            self.calibrate_pending = False
        return ops.rmsnorm(None, self.final_norm, d.eps, delta=delta, residual=resid)

    def set_kv_scales(self, k_scale, v_scale) -> None:
        """Per-layer fp8 KV scales into the native engine(s): the device table the captured decode steps read and the arguments
        of later prefill launches.  Graphs captured before the call pick the new scales up (they read the table)."""
        if self.kv_code != L.KV_FP8:
            raise L.OmniError("set_kv_scales: fp8 KV only")
        n = self.d.layers
        self.k_scale_l, self.v_scale_l = [float(v) for v in k_scale], [float(v) for v in v_scale]
        if len(self.k_scale_l) != n or len(self.v_scale_l) != n:
            raise ValueError(f"set_kv_scales: {n} layers")
        ka, va = (C.c_float * n)(*self.k_scale_l), (C.c_float * n)(*self.v_scale_l)
        for h in [self.handle] + [hk for hk, _, _ in self._sub]:
            L.check(self.lib.omni_talker_set_kv_scales(h, ka, va, L.current_stream()), "omni_talker_set_kv_scales")

    def prefill_blas(self, x, positions, req_of_tok, slot_mapping, block_table=None):
        return self.prefill_wide(x, positions, req_of_tok, slot_mapping, block_table, gemm="blas")

    def _moe_group_slots(self, idx: torch.Tensor, wts: torch.Tensor, T: int):
        """(token, expert) slots grouped by expert into a padded [E, cap, .] batch: returns (order, dst, wts_s, counts, cap) with
        slot order[s] (= t * k + j of the token's j-th expert in ascending expert order) stored at batch row dst[s]; cap = the
        busiest expert's token count rounded to 64 (one host read per layer)."""
        E, k = self.moe_experts_l, self.d.moe_top_k
        if self.moe_mode == "ep":       # slots of experts another rank holds: weight 0 here (their share arrives by all-reduce)
            local = (idx >= self.moe_e0) & (idx < self.moe_e0 + E)
            wts = torch.where(local, wts, torch.zeros_like(wts))
            idx = torch.where(local, idx - self.moe_e0, torch.zeros_like(idx))
        idx_s, perm = torch.sort(idx.long(), dim=1)                     # per token: its experts in ascending order
        wts_s = torch.gather(wts, 1, perm)
        flat_e = idx_s.reshape(-1)                                      # expert of slot (t, j), slot = t * k + j
        order = torch.argsort(flat_e, stable=True)                      # slots grouped by expert
        counts = torch.bincount(flat_e, minlength=E)
        cap = (max(int(counts.max().item()), 1) + 63) // 64 * 64
        e_sorted = flat_e[order]
        rank = torch.arange(T * k, device=idx.device) - (torch.cumsum(counts, 0) - counts)[e_sorted]
        return order, e_sorted * cap + rank, wts_s, counts, cap

    def _moe_combine(self, y: torch.Tensor, order, dst, wts_s, T: int) -> torch.Tensor:
        """Weighted expert outputs added back per token in ascending expert order in bf16 -- the accumulation order and the
        rounding points of HF's expert loop (oracle.moe_block)."""
        k, H = self.d.moe_top_k, y.shape[1]
        ys = torch.empty(T * k, H, dtype=BF16, device=y.device)
        ys[order] = y[dst]
        ys = ys.view(T, k, H) * wts_s[:, :, None]
        out = torch.zeros(T, H, dtype=BF16, device=y.device)
        for j in range(k):
            out += ys[:, j]
        return out

    def _moe_mlp_tile(self, a: torch.Tensor, w: dict) -> torch.Tensor:
        """Sparse-MoE MLP over T prompt rows on the hand-written MFMA kernel: router, both expert GEMMs (ONE grouped
        omni_gemm_tile launch each over the expert-sorted [E, cap, H] batch, reading the decode step's fragment-major expert
        matrices; tiles past an expert's live rows leave at once) and the shared expert -- no library GEMM on the path.
        Same grouping, rounding points and accumulation order as _moe_mlp_blas (reference: vLLM FusedMoE under
        V/model_executor/models/qwen3_omni/qwen3_moe.py:8,152-161; arithmetic = HF's expert loop, oracle.moe_block)."""
        d = self.d
        T, H = a.shape
        E, k, I = self.moe_experts_l, d.moe_top_k, self.moe_inter_l
        idx, wts = ops.moe_route(ops.gemm_tile(a, w["moe_router_f"]), k, d.moe_norm_topk)      # int32 [T, k], bf16 [T, k]
        order, dst, wts_s, counts, cap = self._moe_group_slots(idx, wts, T)
        rows = counts.to(torch.int32)
        xb = torch.empty(E * cap, H, dtype=BF16, device=a.device)       # rows past an expert's count are never read
        xb[dst] = a[order // k]
        gu = ops.gemm_tile(xb, w["moe_gate_up_f"], groups=E, group_rows=rows)                   # [E * cap, 2I]
        act = ops.silu_mul(gu)
        y = ops.gemm_tile(act, w["moe_down_f"], groups=E, group_rows=rows)                      # [E * cap, H]
        out = self._moe_combine(y, order, dst, wts_s, T)
        if self.moe_shared_l > 0:
            sh = ops.gemm_tile(ops.silu_mul(ops.gemm_tile(a, w["moe_shared_gate_up_f"])), w["moe_shared_down_f"])
            gate = ops.gemm_tile(a, w["moe_shared_gate_f"])[:, :1]
            out = out + torch.sigmoid(gate) * sh
        return out

    def _moe_mlp_blas(self, a: torch.Tensor, w: dict) -> torch.Tensor:
        """The same block with the expert GEMMs as TWO batched hipBLASLt calls on [E, K, N] copies (prefill_gemm="blas": the
        A/B arm of _moe_mlp_tile; padded rows are zero)."""
        import torch.nn.functional as F
        d = self.d
        T, H = a.shape
        E, k, I = self.moe_experts_l, d.moe_top_k, self.moe_inter_l
        idx, wts = ops.moe_route(F.linear(a, w["moe_router"]), k, d.moe_norm_topk)       # int32 [T, k], bf16 [T, k]
        order, dst, wts_s, counts, cap = self._moe_group_slots(idx, wts, T)
        xb = torch.zeros(E * cap, H, dtype=BF16, device=a.device)
        xb[dst] = a[order // k]
        gu = torch.bmm(xb.view(E, cap, H), w["moe_gate_up_t"])                            # [E, cap, 2I]
        act = ops.silu_mul(gu.view(E * cap, 2 * I))
        y = torch.bmm(act.view(E, cap, I), w["moe_down_t"]).view(E * cap, H)
        out = self._moe_combine(y, order, dst, wts_s, T)
        if self.moe_shared_l > 0:
            sh = F.linear(ops.silu_mul(F.linear(a, w["moe_shared_gate_up"])), w["moe_shared_down"])
            out = out + torch.sigmoid(F.linear(a, w["moe_shared_gate"])) * sh
        return out

    def prefill(self, x: torch.Tensor, positions: torch.Tensor, req_of_tok: torch.Tensor, slot_mapping: torch.Tensor,
                block_table: torch.Tensor | None = None, use_blas: bool | None = None, gemm: str | None = None,
                rope_positions: torch.Tensor | None = None) -> torch.Tensor:
        """Backbone over T prompt tokens (x bf16 [T,H]) -> final-normed hidden [T,H].  use_blas (historic name) selects the
        all-tokens-per-layer pass (prefill_wide; its GEMMs per `gemm` / the engine's prefill_gemm) over max_batch-row chunks
        on the decode kernels; default: wide when T > max_batch.  rope_positions int32 [3, T]: M-RoPE ids with differing rows
        (always the all-tokens pass; the decode rows then need engine.rope_delta[row] = the request's mrope_position_delta)."""
        bt = self.block_table if block_table is None else block_table
        T = x.shape[0]
        if use_blas is None:
            use_blas = T > self.max_batch
        if use_blas or rope_positions is not None or self.calibrate_pending:      # the calibration pass is the all-tokens pass
            return self.prefill_wide(x, positions, req_of_tok, slot_mapping, bt, gemm=gemm, rope_positions=rope_positions)
        out = torch.empty_like(x)
        st = L.current_stream()
        if not self.tp_path:
            L.check(self.lib.omni_talker_prefill(self.handle, L.ptr(x), L.ptr(positions), L.ptr(req_of_tok),
                                                 L.ptr(slot_mapping), L.ptr(bt), L.ptr(out), T, st), "omni_talker_prefill")
            return out
        import torch.distributed as dist
        Bm = self.max_batch
        for t0 in range(0, T, Bm):
            rows = min(Bm, T - t0)
            L.check(self.lib.omni_talker_rows_begin(self.handle, x[t0:].data_ptr(), rows, st), "rows_begin")
            for l in range(self.d.layers):
                L.check(self.lib.omni_talker_rows_attn(self.handle, l, rows, positions[t0:].data_ptr(),
                                                       slot_mapping[t0:].data_ptr(), L.ptr(bt), req_of_tok[t0:].data_ptr(), st),
                        "rows_attn")
                dist.all_reduce(self._attn_out[:rows], group=self.tp_group)
                L.check(self.lib.omni_talker_rows_mlp(self.handle, l, rows, st), "rows_mlp")
                dist.all_reduce(self._mlp_out[:rows], group=self.tp_group)
            L.check(self.lib.omni_talker_rows_end(self.handle, out[t0:].data_ptr(), rows, st), "rows_end")
        return out

    def compute_logits(self, hidden: torch.Tensor, round_bf16: bool = True) -> torch.Tensor:
        """lm_head + codec mask (qwen3_tts_talker.py:424-443) -> fp32 [R, vocab]."""
        R = hidden.shape[0]
        out = torch.empty(R, self.d.vocab, dtype=torch.float32, device=self.device)
        L.check(self.lib.omni_talker_logits(self.handle, L.ptr(hidden), L.ptr(out), R, int(round_bf16),
                                            L.current_stream()), "omni_talker_logits")
        return out

    def sample(self, logits: torch.Tensor, *, greedy, temperature=1.0, top_k=0, top_p=1.0, rep_penalty=1.0, seen=None, seed=0,
               steps=None) -> torch.Tensor:
        """Sampler on arbitrary logits rows (first token after prefill); marks `seen`, increments `steps`."""
        return ops.sample(logits, greedy=greedy, temperature=temperature, top_k=top_k, top_p=top_p, rep_penalty=rep_penalty, seen=seen,
                          seed=seed, steps=steps, inc_steps=steps is not None)

    def chain_error(self, reset: bool | int = False) -> int:
        """Sticky error word of the persistent code-predictor launches (0 = no flag wait ever timed out); synchronises.
        reset = 2: fault injection (sets the word as a timed-out wait would)."""
        rc = self.lib.omni_talker_chain_error(self.handle, int(reset))
        if rc < 0:
            L.check(rc, "omni_talker_chain_error")
        for h, _, _ in self._sub:                   # the concurrent row ranges' engines (half-grid backbone chains)
            r2 = self.lib.omni_talker_chain_error(h, int(reset))
            if r2 < 0:
                L.check(r2, "omni_talker_chain_error(sub)")
            rc = rc or r2
        return rc

    def chains_ran(self) -> int:
        """Which persistent chains the last native decode-step call launched (bit 0: code predictor, bit 1: backbone): 0 on the
        launch-per-op path -- other shapes, batches below the backbone chain's row range, tensor-parallel ranks, chains off."""
        ran = int(self.lib.omni_talker_chains_ran(self.handle))
        for h, _, _ in self._sub:
            ran |= int(self.lib.omni_talker_chains_ran(h))
        return ran

    def set_chains(self, on) -> None:
        """Turn the persistent chains of this engine on / off (same bits either way); ``on == 2``: the half grid -- the backbone chain on
        128 workgroups (two engines' launches fit the chip side by side), the code predictor launch per op.  Captured graphs keep what they
        recorded: re-capture after a switch.  ``on == 3``: chains on, a tensor-parallel rank's backbone on all-reduce LAUNCHES."""
        L.check(self.lib.omni_talker_set_chains(self.handle, int(on) if on in (2, 3) and on is not True else int(bool(on))), "omni_talker_set_chains")
        for h, _, _ in self._sub:                   # concurrent row ranges: the half grid or nothing
            L.check(self.lib.omni_talker_set_chains(h, 2 if on == 2 and on is not True and self.n_sub == 2 else 0), "omni_talker_set_chains(sub)")
        self.persistent_chains = bool(on)

    def recover_from_chain_timeout(self) -> None:
        """A flag wait of a persistent chain ran out (status word 0 of a step != 0: the 256-workgroup grid was not co-resident
        -- another process or engine held part of the GPU).  Clear the sticky word and the flags and leave the chains off for
        this engine: its later steps run launch-per-op, which shares a GPU without deadline.  The step that reported the
        word is invalid: the caller redoes it (runner.MI355XARModelRunner._redo_after_chain_timeout)."""
        self.chain_error(reset=True)
        self.set_chains(False)
        self.status.zero_()

    def check_device_errors(self) -> None:
        """The bounded spins of the in-kernel hand-offs (peer all-reduce flags across ranks, stage flags of the persistent
        chains) record a time-out in a sticky device word and carry on with wrong data: whoever consumes step outputs must
        look at those words.  Raises RuntimeError -- the reference's convention for a dead stage (engine core
        sees the exception) -- instead of handing wrong audio on."""
        err = self.chain_error()
        if err:
            raise RuntimeError(f"persistent-chain flag wait timed out (stage code {err:#x}): the step's outputs are invalid")
        if self.ar is not None and self.ar.error() != 0:
            raise RuntimeError(f"peer all-reduce: a rank did not arrive in time (error word {self.ar.error()}): the step's outputs are invalid")

    def sample_rows(self, logits: torch.Tensor, rows: torch.Tensor, *, seen=None, steps=None) -> torch.Tensor:
        """First token after prefill for batch rows `rows` (int64 device indices), each with its own request's parameters."""
        par = {k: getattr(self, "row_" + k).index_select(0, rows) for k in self.ROW_KEYS}
        return ops.sample_rows(logits, par, seen=seen, steps=steps, inc_steps=steps is not None)

    def code_predictor(self, layer0_ids, layer0_embed, last_hidden, *, greedy=True, temperature=0.9, top_k=50, top_p=1.0, seed=0,
                       steps=None, return_logits=False):
        B = layer0_ids.shape[0]
        Q = self.d.num_code_groups
        codes = torch.empty(B, Q, dtype=torch.int64, device=self.device)
        lg = torch.empty(B, Q - 1, self.d.codebook, dtype=torch.float32, device=self.device) if return_logits else None
        L.check(self.lib.omni_talker_code_predictor(
            self.handle, L.ptr(layer0_ids), L.ptr(layer0_embed), L.ptr(last_hidden), L.ptr(codes), L.ptr(lg), B,
            int(greedy), float(temperature), int(top_k), float(top_p), int(seed) & 0xFFFFFFFF, L.ptr(steps), L.current_stream()),
            "omni_talker_code_predictor")
        return (codes, lg) if return_logits else codes

    # ------------------------------------------------------------------ roofline accounting
    def step_bytes(self, ctx_lens) -> dict:
        """Algorithmic HBM bytes of one decode step on this rank (each counted once)."""
        from .weights import weight_bytes
        d = self.d
        wb = weight_bytes(d)
        kvb = 2 if self.kv_code in (L.KV_BF16, L.KV_FP16) else 1
        per_tok = d.layers * 2 * self.hkv_l * d.head_dim * kvb
        if self.kv_code == L.KV_INT8:
            per_tok += d.layers * 2 * self.hkv_l * 4
        kv_rd = int(sum(int(c) for c in ctx_lens)) * per_tok
        kv_wr = len(ctx_lens) * per_tok
        return {"weights_backbone": wb["backbone"] // self.tp_size, "lm_head": wb["lm_head"],
                "code_predictor": wb["code_predictor"], "kv_read": kv_rd, "kv_write": kv_wr,
                "total": wb["backbone"] // self.tp_size + wb["lm_head"] + wb["code_predictor"] + kv_rd + kv_wr}
