"""Time the HIP log-VQT front end on a song-length waveform.
usage: python tools/bench_vqt.py [seconds_of_audio]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from osufusion_amd import audio as A  # noqa: E402


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
    n = int(secs * A.SR)
    y = (torch.randn(n, generator=torch.Generator().manual_seed(0)) * 0.1)
    yd = y.cuda()
    for _ in range(3):
        out = A.log_vqt(yd)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    reps = 20
    ev[0].record()
    for _ in range(reps):
        out = A.log_vqt(yd)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    frames = out.shape[1]
    K = A.vqt_bank().bank.shape[1]
    flops = 2.0 * frames * K * 192
    print(f"audio {secs:.0f} s -> {frames} frames: {ms:.3f} ms per song ({secs / (ms * 1e-3):.0f}x real time), "
          f"GEMM {flops / (ms * 1e-3) / 1e12:.1f} TFLOP/s fp32-equivalent incl. padding + log/transpose")
    t0 = time.perf_counter()
    y.cpu()
    A.log_vqt(y)                                          # host waveform: includes the PCIe upload
    torch.cuda.synchronize()
    print(f"from a host waveform (upload included): {(time.perf_counter() - t0) * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
