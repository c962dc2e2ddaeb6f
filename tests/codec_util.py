"""Seeded weights of the Code2Wav (12 Hz tokenizer) decoder under the reference's parameter names, shared by the fixture
maker (tests/golden/make_fixtures.py loads them into the reference's Qwen3TTSTokenizerV2Decoder) and the tests (which hand
the same state dict to the oracle and to the product decoder).  Data, not model code: names and shapes follow
tokenizer_12hz/modeling_qwen3_tts_tokenizer_v2.py:912-960; the values are drawn so that every part of the network matters
(layer scales / ConvNeXt gammas of order 0.5 rather than the 0.01 / 1e-6 initialisers, non-zero codebooks)."""
from __future__ import annotations

import math

import torch

# tiny: the reference module runs it on CPU in milliseconds; every channel count a multiple of 32 (the GEMM kernel's k-step)
TINY_CODEC = dict(codebook_size=64, codebook_dim=64, hidden_size=128, latent_dim=64, num_attention_heads=2, num_key_value_heads=2,
                  sliding_window=6, intermediate_size=256, num_hidden_layers=2, num_quantizers=4, upsample_rates=(4, 3),
                  upsampling_ratios=(2, 2), decoder_dim=128, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=256)
# mid: the real architecture's depth (8 transformer layers, 4 decoder blocks, window 72) at a quarter of its widths
MID_CODEC = dict(codebook_size=256, codebook_dim=128, hidden_size=256, latent_dim=256, num_attention_heads=4, num_key_value_heads=4,
                 sliding_window=72, intermediate_size=768, num_hidden_layers=8, num_quantizers=16, upsample_rates=(8, 5, 4, 3),
                 upsampling_ratios=(2, 2), decoder_dim=512, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=8000)
# full: Qwen3TTSTokenizerV2DecoderConfig defaults (configuration_qwen3_tts_tokenizer_v2.py:74-95) + codebook_dim 512
FULL_CODEC = dict(codebook_size=2048, codebook_dim=512, hidden_size=1024, latent_dim=1024, num_attention_heads=16, num_key_value_heads=16,
                  sliding_window=72, intermediate_size=3072, num_hidden_layers=8, num_quantizers=16, upsample_rates=(8, 5, 4, 3),
                  upsampling_ratios=(2, 2), decoder_dim=1536, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=8000)


def make_codec_state(cfg: dict, seed: int = 0, device: str = "cpu") -> dict[str, torch.Tensor]:
    g = torch.Generator(device=device).manual_seed(seed)
    sd: dict[str, torch.Tensor] = {}

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, device=device) * std

    def lin(name, out_f, in_f, bias=True, gain=1.0):
        sd[name + ".weight"] = rn(out_f, in_f, std=gain / math.sqrt(in_f))
        if bias:
            sd[name + ".bias"] = rn(out_f, std=0.02)

    def conv(name, out_c, in_c, k, gain=1.0):
        sd[name + ".conv.weight"] = rn(out_c, in_c, k, std=gain / math.sqrt(in_c * k))
        sd[name + ".conv.bias"] = rn(out_c, std=0.02)

    def tconv(name, in_c, out_c, k, stride):
        sd[name + ".conv.weight"] = rn(in_c, out_c, k, std=1.0 / math.sqrt(in_c * k / stride))
        sd[name + ".conv.bias"] = rn(out_c, std=0.02)

    def snake(name, c):
        sd[name + ".alpha"] = rn(c, std=0.3)
        sd[name + ".beta"] = rn(c, std=0.3)

    H, Lt, I = cfg["hidden_size"], cfg["latent_dim"], cfg["intermediate_size"]
    hd = H // cfg["num_attention_heads"]
    cd, bins, nq = cfg["codebook_dim"], cfg["codebook_size"], cfg["num_quantizers"]
    # ---- pre_transformer
    for l in range(cfg["num_hidden_layers"]):
        p = f"pre_transformer.layers.{l}."
        lin(p + "self_attn.q_proj", cfg["num_attention_heads"] * hd, H, bias=False)
        lin(p + "self_attn.k_proj", cfg["num_key_value_heads"] * hd, H, bias=False)
        lin(p + "self_attn.v_proj", cfg["num_key_value_heads"] * hd, H, bias=False)
        lin(p + "self_attn.o_proj", H, cfg["num_attention_heads"] * hd, bias=False)
        lin(p + "mlp.gate_proj", I, H, bias=False)
        lin(p + "mlp.up_proj", I, H, bias=False)
        lin(p + "mlp.down_proj", H, I, bias=False)
        sd[p + "input_layernorm.weight"] = 1 + rn(H, std=0.1)
        sd[p + "post_attention_layernorm.weight"] = 1 + rn(H, std=0.1)
        sd[p + "self_attn_layer_scale.scale"] = 0.5 + rn(H, std=0.1)
        sd[p + "mlp_layer_scale.scale"] = 0.5 + rn(H, std=0.1)
    sd["pre_transformer.norm.weight"] = 1 + rn(H, std=0.1)
    lin("pre_transformer.input_proj", H, Lt)
    lin("pre_transformer.output_proj", Lt, H)
    # ---- quantizer (SplitResidualVectorQuantizer: 1 semantic + nq - 1 acoustic codebooks of dimension codebook_dim / 2)
    for part, n in (("rvq_first", 1), ("rvq_rest", nq - 1)):
        sd[f"quantizer.{part}.input_proj.weight"] = rn(cd // 2, cd, 1, std=1 / math.sqrt(cd))          # unused by decode
        sd[f"quantizer.{part}.output_proj.weight"] = rn(cd, cd // 2, 1, std=1 / math.sqrt(cd // 2))
        for q in range(n):
            sd[f"quantizer.{part}.vq.layers.{q}._codebook.cluster_usage"] = torch.rand(bins, generator=g, device=device) * 1.5 + 0.5
            sd[f"quantizer.{part}.vq.layers.{q}._codebook.embedding_sum"] = rn(bins, cd // 2, std=1.0 / math.sqrt(n))
    conv("pre_conv", Lt, cd, 3)
    # ---- upsample: (transposed conv factor f, ConvNeXt) per ratio
    for i, f in enumerate(cfg["upsampling_ratios"]):
        tconv(f"upsample.{i}.0", Lt, Lt, f, f)
        p = f"upsample.{i}.1."
        sd[p + "dwconv.conv.weight"] = rn(Lt, 1, 7, std=1 / math.sqrt(7))
        sd[p + "dwconv.conv.bias"] = rn(Lt, std=0.02)
        sd[p + "norm.weight"] = 1 + rn(Lt, std=0.1)
        sd[p + "norm.bias"] = rn(Lt, std=0.05)
        lin(p + "pwconv1", 4 * Lt, Lt)
        lin(p + "pwconv2", Lt, 4 * Lt)
        sd[p + "gamma"] = 0.5 + rn(Lt, std=0.1)
    # ---- decoder: conv7, blocks, snake, conv7 -> 1
    D = cfg["decoder_dim"]
    conv("decoder.0", D, Lt, 7)
    for i, r in enumerate(cfg["upsample_rates"]):
        cin, cout = D // 2 ** i, D // 2 ** (i + 1)
        p = f"decoder.{i + 1}.block."
        snake(p + "0", cin)
        tconv(p + "1", cin, cout, 2 * r, r)
        for u in range(3):
            q = f"{p}{u + 2}."
            snake(q + "act1", cout)
            conv(q + "conv1", cout, cout, 7, gain=0.5)
            snake(q + "act2", cout)
            conv(q + "conv2", cout, cout, 1, gain=0.3)
    n = len(cfg["upsample_rates"])
    cl = D // 2 ** n
    snake(f"decoder.{n + 1}", cl)
    conv(f"decoder.{n + 2}", 1, cl, 7, gain=0.12)
    return sd


def total_upsample(cfg: dict) -> int:
    return math.prod(cfg["upsample_rates"]) * math.prod(cfg["upsampling_ratios"])
