"""How fast does the host read a pinned staging buffer the GPU has just written (D2H on a side stream)?  Decides the form of the
async output path's staging (runner.AsyncStepOutput)."""
import time, torch, numpy as np
dev = torch.device("cuda:0")
src = torch.randn(64, 2048, device=dev).to(torch.bfloat16)
ids = torch.arange(68, dtype=torch.int32, device=dev)
codes = torch.arange(64 * 16, dtype=torch.int64, device=dev).reshape(64, 16)
pin_h = torch.empty(64, 2048, dtype=torch.bfloat16).pin_memory()
pin_i = torch.empty(68, dtype=torch.int32).pin_memory()
pin_c = torch.empty(64, 16, dtype=torch.int64).pin_memory()
side = torch.cuda.Stream()
def t(f, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def pageable():
    a = ids.cpu(); b = src.to("cpu", copy=True); c = codes.cpu(); return a.tolist(), c.tolist()
def pinned_async():
    ev = torch.cuda.Event()
    with torch.cuda.stream(side):
        pin_i.copy_(ids, non_blocking=True); pin_h.copy_(src, non_blocking=True); pin_c.copy_(codes, non_blocking=True)
        ev.record(side)
    ev.synchronize()
def pinned_read_tolist():
    pinned_async(); return pin_i.tolist(), pin_c.tolist()
def pinned_read_clone():
    pinned_async(); return pin_i.clone().tolist(), pin_c.clone().tolist(), pin_h.clone()
def pinned_read_numpy():
    pinned_async(); return pin_i.numpy().copy().tolist(), pin_c.numpy().copy().tolist(), pin_h.view(torch.int16).numpy().copy()
print(f"pageable 3 x .cpu() + tolist      {t(pageable):8.1f} us")
print(f"pinned async copies + event sync  {t(pinned_async):8.1f} us")
print(f"  + tolist straight off pinned    {t(pinned_read_tolist):8.1f} us")
print(f"  + clone() then tolist, hidden   {t(pinned_read_clone):8.1f} us")
print(f"  + numpy copy                    {t(pinned_read_numpy):8.1f} us")
x = torch.empty(64, 2048, dtype=torch.bfloat16)
print(f"plain host clone of 256 KB        {t(lambda: x.clone()):8.1f} us")
print(f"pinned host clone of 256 KB       {t(lambda: pin_h.clone()):8.1f} us")
idx = np.arange(64)
pin_idx = torch.empty(64, dtype=torch.int64).pin_memory()
d_idx = torch.empty(64, dtype=torch.int64, device=dev)
def h2d_pageable(): return torch.as_tensor(idx, device=dev)
def h2d_pinned(): pin_idx.numpy()[:] = idx; d_idx.copy_(pin_idx, non_blocking=True)
print(f"H2D 64 idx pageable as_tensor     {t(h2d_pageable):8.1f} us")
print(f"H2D 64 idx pinned non_blocking    {t(h2d_pinned):8.1f} us")
# does a pageable H2D block the host until earlier work on the stream is done?
big = torch.randn(8192, 8192, device=dev)
def busy_then(f):
    torch.cuda.synchronize()
    for _ in range(20): big @ big
    t0 = time.perf_counter(); f(); dt = time.perf_counter() - t0
    torch.cuda.synchronize(); return dt * 1e6
print(f"host time of pageable H2D behind ~20 matmuls  {busy_then(h2d_pageable):8.1f} us (blocks if large)")
print(f"host time of pinned   H2D behind ~20 matmuls  {busy_then(h2d_pinned):8.1f} us")
print(f"host time of fill_ behind ~20 matmuls          {busy_then(lambda: d_idx.fill_(3)):8.1f} us")
