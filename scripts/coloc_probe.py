"""Determinism probe behind tests/test_gpu_colocation.py: the same 500 captured steps solo twice, then beside the Code2Wav child;
reports the first step / row / group where the codes differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp
from tests.test_gpu_colocation import _replay, _code2wav_loop
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.weights import make_weights


def first_diff(a, b, name):
    if torch.equal(a, b):
        print(name, "identical")
        return
    d = (a != b).nonzero()
    s, r, g = d[0].tolist()
    print(name, "DIFFER: first at step", s, "row", r, "group", g, "| differing (step,row) pairs in that step:",
          int((a[s] != b[s]).any(-1).sum()), "| groups differing at that cell:", (a[s, r] != b[s, r]).nonzero().flatten().tolist())


def main():
    d = get_dims("tts-1.7b").with_(layers=4, max_model_len=1024)
    w = make_weights(d, seed=4, std=0.02)
    B, steps = 64, int(os.environ.get("STEPS", 500))
    s1 = _replay(d, w, B, steps)[0]
    s2 = _replay(d, w, B, steps)[0]
    first_diff(s1, s2, "solo vs solo")
    ctx = mp.get_context("spawn")
    ready, stop, count = ctx.Event(), ctx.Event(), ctx.Value("i", 0)
    child = ctx.Process(target=_code2wav_loop, args=(ready, stop, count))
    child.start()
    try:
        assert ready.wait(300)
        c1, st, ms, err, ran = _replay(d, w, B, steps)
        c2 = _replay(d, w, B, steps)[0]
    finally:
        stop.set()
        child.join(120)
    print("err", hex(err), "ran", ran, "status sum", int(st[:, :2].abs().sum()))
    first_diff(s1, c1, "solo vs colocated")
    first_diff(c1, c2, "colocated vs colocated")


if __name__ == "__main__":
    main()
