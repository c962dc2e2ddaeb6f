"""Wall time per engine step THROUGH the reference's loop -- scheduler.schedule -> worker.execute_model -> sample_tokens (-> AsyncStepOutput
.get_output) -> scheduler.update_from_output -- W3 shape, B = 64, next to bench.py's bare hipGraph replay: the same measurement as the
bench line's `engine_loop` key (bench.engine_loop), async scheduling on and off, for quick runs on a GPU box.
usage: python scripts/bench_runner.py [--steps 200] [--batch 64] [--kv fp8]"""
import argparse, json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ht_vllm_omni_amd.config import get_dims
from ht_vllm_omni_amd.weights import make_weights

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--kv", default="fp8")
ap.add_argument("--model", default="tts-1.7b")
a = ap.parse_args()
torch.cuda.set_device(0)
d = get_dims(a.model)
w = make_weights(d, seed=1234, std=0.02)
args = types.SimpleNamespace(batch=a.batch, kv=a.kv, num_blocks=8192, target_ctx=352)
lens = np.random.default_rng(7).integers(32, 161, size=a.batch).tolist()
for mode in (True, False):
    r = bench.engine_loop(d, w, args, lens, mode, n_steps=a.steps)
    print(("async " if mode else "sync  ") + json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))
