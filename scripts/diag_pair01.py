import os as _os; _os.environ.setdefault("OMNI_TALKER_DEBUG", "1")   # omni_debug_* hooks live in libomni_talker_debug.so
import sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from tests.test_gpu_engine import _engine, get_dims, make_weights, O, BF16
from ht_vllm_omni_amd import _lib as L
lib = L.load(); lib.omni_debug_cp_pair01.argtypes=[C.c_int]; lib.omni_debug_cp_pair01.restype=None
for seed in (13, 14, 15):
    d = get_dims("tts-0.6b").with_(layers=1, cp_layers=2, max_model_len=256)
    w = make_weights(d, seed=seed, std=0.02)
    B = 12
    orc = O.TalkerOracle(d, w)
    g = torch.Generator().manual_seed(B)
    code0 = torch.randint(1, d.codebook, (B,), generator=g)
    e0 = w["embed"][code0]; lh = torch.randn(B, d.hidden, generator=g).to(BF16)
    ref_codes, ref_lg = orc.code_predictor(code0, e0, lh, do_sample=False, return_logits=True)
    for on in (0, 1):
        lib.omni_debug_cp_pair01(on)
        eng = _engine(d, w, kv_dtype="bf16", num_blocks=8, max_batch=16)
        codes, lg = eng.code_predictor(code0.to(torch.int32).cuda(), e0.cuda(), lh.cuda(), greedy=True, return_logits=True)
        dd = (lg.cpu()[:, 0].float() - ref_lg[:, 0].float()).abs()
        print("seed", seed, "pair", on, "mean %.5f max %.4f" % (dd.mean().item(), dd.max().item()), "codes eq", (codes.cpu()[:,1]==ref_codes[:,1]).float().mean().item())
