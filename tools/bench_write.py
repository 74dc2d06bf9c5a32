import sys
sys.path.insert(0, "/root/repo")
import torch
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for mb in (67, 302, 1200):
    n = mb * 1000 * 1000 // 2
    a = torch.empty(n, device="cuda", dtype=torch.bfloat16); b = torch.empty_like(a)
    t = timeit(lambda: a.fill_(1.0)); print(f"fill  {mb} MB: {t*1e3:.1f} us  {mb/t/1e3:.2f} TB/s written")
    t = timeit(lambda: a.zero_()); print(f"zero  {mb} MB: {t*1e3:.1f} us  {mb/t/1e3:.2f} TB/s written")
    t = timeit(lambda: b.copy_(a)); print(f"copy  {mb} MB: {t*1e3:.1f} us  {2*mb/t/1e3:.2f} TB/s r+w")
    t = timeit(lambda: a.sum()); print(f"read  {mb} MB: {t*1e3:.1f} us  {mb/t/1e3:.2f} TB/s read")
