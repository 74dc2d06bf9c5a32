"""Does running a layer's input-gradient GEMM (NT) and its weight-gradient GEMM (TN) on TWO streams pay?  Both depend only on dY; on one stream
they serialise, each with its own partial tile round, pipeline fill and write-out tail.   python tools/overlap_probe.py"""
import sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
def run(nt, tn, two, iters):
    for _ in range(iters):
        if two:
            ev = torch.cuda.Event(); ev.record(main)
            nt()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                tn()
            main.wait_stream(side)
        else:
            nt(); tn()
def timeit(nt, tn, two, iters=20):
    run(nt, tn, two, 3); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); run(nt, tn, two, iters); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B = 32
for L, Cin, Cout, taps in ((4096, 256, 256, 3), (2048, 512, 512, 3), (1024, 768, 768, 3), (512, 1024, 1024, 3), (4096, 256, 1152, 1), (4096, 1024, 256, 1), (4096, 256, 1024, 1), (2048, 512, 2048, 1)):
    M = B * L
    dy = torch.randn(M, Cout, device="cuda").bfloat16(); x = torch.randn(M, Cin, device="cuda").bfloat16()
    wd = (torch.randn(taps, Cin, Cout, device="cuda") * 0.05).bfloat16()           # dgrad operand: [tap'][C_in][C_out]
    dx = torch.empty(M, Cin, device="cuda", dtype=torch.bfloat16); dw = torch.empty(taps, Cout, Cin, device="cuda")
    if taps == 1:
        nt = lambda: ops.gemm_nt(dy, wd, None, out=dx)
        tn = lambda: ops.gemm_tn(dy, x, out=dw)
    else:
        nt = lambda: ops.gemm_nt(dy, wd, None, taps=taps, lin=L, lout=L, stride=1, pad=1, out=dx)
        tn = lambda: ops.gemm_tn(dy, x, taps=taps, lin=L, lout=L, stride=1, pad=1, out=dw)
    t_nt = timeit(nt, lambda: None, False); t_tn = timeit(lambda: None, tn, False)
    res = []
    for rnd in range(3):
        res.append((timeit(nt, tn, False), timeit(nt, tn, True)))
    one = min(r[0] for r in res); two = min(r[1] for r in res)
    print(f"B*L={M:6d} {Cin:4d}->{Cout:4d} k{taps}: dgrad {t_nt:6.1f} us  wgrad {t_tn:6.1f} us  one stream {one:6.1f} us  two streams {two:6.1f} us  ({one / two:4.2f}x)", flush=True)
