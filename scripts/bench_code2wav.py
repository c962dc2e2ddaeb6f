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
    # ---- roofline of the dominant kernel (the fused residual unit of the 96-channel block at the last window's row count):
    # algorithmic bytes = s in (2 B) + h in / out (4 + 4 B) + s_next out (2 B) per element + the weights once; HIP events over
    # back-to-back launches on this stream
    from ht_vllm_omni_amd import _lib as L
    import ctypes as C
    lib = L.load()
    T = Ts[-1] * dec.total_upsample
    Cc = 96
    blk = dec.blocks[-1]
    un = blk["units"][1]
    s_in = torch.randn(T, Cc, device="cuda").to(torch.bfloat16)
    s_out = torch.empty_like(s_in)
    h = torch.randn(T, Cc, device="cuda")
    ru = L.ResUnit()
    ru.s, ru.h, ru.s_next = s_in.data_ptr(), h.data_ptr(), s_out.data_ptr()
    ru.w1, ru.b1, ru.snake2_alpha, ru.snake2_inv_beta = un["conv1"].w.data_ptr(), un["conv1"].bias.data_ptr(), un["act2"][0].data_ptr(), un["act2"][1].data_ptr()
    ru.w2, ru.b2, ru.next_alpha, ru.next_inv_beta = un["conv2"].w.data_ptr(), un["conv2"].bias.data_ptr(), un["act1"][0].data_ptr(), un["act1"][1].data_ptr()
    ru.T, ru.C, ru.dilation = T, Cc, un["conv1"].dilation
    t_unit = timed(lambda: L.check(lib.omni_codec_res_unit(C.byref(ru), L.current_stream())), 50)
    alg = T * Cc * 12 + (7 * Cc * Cc + Cc * Cc) * 2
    roof = {"kernel": "res_unit_kernel<96>", "bound": "hbm", "achieved": alg / t_unit * 1e-9, "peak": 8000.0, "unit": "GB/s",
            "frac": alg / t_unit * 1e-9 / 8000.0, "rows": T, "us_per_launch": t_unit * 1e6, "algorithmic_bytes": alg, "traffic": None,
            "whole_decode_mfma": {"achieved": rows[-1]["gemm_tflops"], "peak": 2500.0, "unit": "TFLOP/s", "frac": rows[-1]["gemm_tflops"] / 2500.0}}
    tf = sorted(p for p in os.listdir("profiles") if p.startswith("r") and p.endswith("_pmc_res_unit_traffic.json")) if os.path.isdir("profiles") else []
    if tf:
        roof["traffic"] = json.load(open(os.path.join("profiles", tf[-1])))["traffic_bytes_per_launch"]
    # ---- CPU baseline: the oracle (fp32 torch ops on the host cores) on a bounded sample: one 25-frame window
    cpu = None
    if os.environ.get("CPU_LEG", "1") == "1":
        from oracle.code2wav_oracle import Code2WavOracle
        sd_cpu = {k: v.cpu() for k, v in sd.items()}
        orc = Code2WavOracle(FULL_CODEC, sd_cpu)
        cc = torch.randint(0, 2048, (1, 16, 25))
        t0 = time.perf_counter()
        with torch.no_grad():
            orc(cc)
        dt = time.perf_counter() - t0
        cpu = {"value": 25 * dec.total_upsample / 24000.0 / dt, "unit": "audio seconds per second", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"one 25-frame window (2 s of audio) through oracle/code2wav_oracle.py in fp32: {dt:.2f} s"}
    out = {"metric": "code2wav audio seconds per second (real-time factor)", "value": rows[-1]["rtf_x"], "unit": "x real time", "n_gpus": 1,
           "dtype": "bf16 operands, fp32 accumulate / residual streams", "data": "synthetic",
           "config": {"workload": f"Qwen3-TTS-Tokenizer-12Hz decoder architecture, random-init weights, {Ts[-1]}-frame window, hipGraph replay"},
           "windows": rows, "roofline": roof, "cpu_baseline": cpu}
    print(json.dumps(out))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/code2wav_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
