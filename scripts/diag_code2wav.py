"""Per-stage deviation of the HIP Code2Wav decoder from the fp32 oracle and from the oracle with bf16 rounding points (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle.code2wav_oracle import Code2WavOracle
from tests.codec_util import MID_CODEC, TINY_CODEC, make_codec_state
from ht_vllm_omni_amd.code2wav import Code2WavDecoder

cfg = {"mid": MID_CODEC, "tiny": TINY_CODEC}[sys.argv[1] if len(sys.argv) > 1 else "mid"]
T = int(sys.argv[2]) if len(sys.argv) > 2 else 13
sd = make_codec_state(cfg, 3)
dec = Code2WavDecoder(cfg, sd)
codes = torch.randint(0, cfg["codebook_size"], (1, cfg["num_quantizers"], T), generator=torch.Generator().manual_seed(T))
taps, o32, o16 = {}, {}, {}
wav = dec.forward(codes.cuda(), taps).cpu()
w32 = Code2WavOracle(cfg, sd).forward(codes, o32)
w16 = Code2WavOracle(cfg, sd, bf16_points=True).forward(codes, o16)
for k in taps:
    g = taps[k].float().cpu()
    for name, o in (("fp32", o32), ("bf16pts", o16)):
        r = o[k]
        r = r if r.shape == g.shape else r.T
        d = (g - r).abs()
        print(f"{k:16s} vs {name:8s} |ref| mean {r.abs().mean():8.4f} max {r.abs().max():8.3f}   rel mean diff {d.mean() / r.abs().mean():.5f}  max diff {d.max():.4f}")
for name, w in (("fp32", w32), ("bf16pts", w16)):
    d = (wav - w).abs()
    print(f"wav vs {name:8s} |ref| mean {w.abs().mean():.4f}  mean diff {d.mean():.5f}  max diff {d.max():.4f}")
d = (w16 - w32).abs(); print(f"bf16pts vs fp32 oracle: mean {d.mean():.5f} max {d.max():.4f}")
