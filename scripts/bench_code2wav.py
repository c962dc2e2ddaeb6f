"""Code2Wav decoder at the full Qwen3-TTS-Tokenizer-12Hz architecture (random-init weights): time per decode of a window of T
code frames, eager and under hipGraph replay; real-time factor (audio seconds per wall second), GEMM-shaped TFLOP/s, and the
same network as fp32 / bf16 torch modules (MIOpen / hipBLASLt: what the reference's decoder runs on ROCm) timed beside it."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.codec_util import FULL_CODEC, make_codec_state
from ht_vllm_omni_amd.code2wav import Code2WavDecoder


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def torch_decoder(sd, dtype):
    """The oracle's functional restatement moved to the GPU (torch ops on MIOpen / hipBLASLt, channel-major like the reference)."""
    from oracle.code2wav_oracle import Code2WavOracle
    o = Code2WavOracle(FULL_CODEC, {})
    o.sd = {k: v.to("cuda", dtype) for k, v in sd.items()}
    return o


def main():
    Ts = [int(t) for t in (sys.argv[1:] or ["50", "325"])]
    sd = make_codec_state(FULL_CODEC, 0, device="cuda")
    dec = Code2WavDecoder(FULL_CODEC, sd)
    rows = []
    for T in Ts:
        codes = torch.randint(0, 2048, (1, 16, T), device="cuda")
        eager = timed(lambda: dec(codes), 10)
        g = torch.cuda.CUDAGraph()
        dec(codes); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            out = dec(codes)
        graph = timed(g.replay, 20)
        audio_s = T * dec.total_upsample / 24000.0
        row = {"frames": T, "audio_s": audio_s, "eager_ms": eager * 1e3, "graph_ms": graph * 1e3, "rtf_x": audio_s / graph,
               "gemm_tflops": dec.flops(T) / graph * 1e-12, "gflop": dec.flops(T) * 1e-9}
        if os.environ.get("TORCH_LEG", "1") == "1":
            for name, dt in (("torch_fp32_ms", torch.float32), ("torch_bf16_ms", torch.bfloat16)):
                try:
                    o = torch_decoder(sd, dt)
                    with torch.no_grad():
                        row[name] = timed(lambda: o.forward(codes), 3) * 1e3
                except Exception as e:   # noqa: BLE001
                    row[name] = f"failed: {type(e).__name__}: {e}"[:120]
        rows.append(row)
        print(json.dumps(row))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(rows, open("gpurun_out/code2wav_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
