"""What a write-only / read-only stream reaches on this MI355X (torch fill_ / sum over 1.68 GB = conv_first_k's output for sixteen
640 x 640 frames, float4 per lane), next to the copy rate: the ceilings the two thin kernels are measured against in DESIGN.md §4.
    python tools/hbm_ceiling.py"""
import time
import torch
n = 16 * 642 * 642 * 64
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
def rate(fn, bytes_moved, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return bytes_moved * reps / (time.perf_counter() - t0) / 1e12
print("write only (fill_):   %.2f TB/s" % rate(lambda: x.fill_(1.5), 4 * n))
print("hipMemsetAsync zero_: %.2f TB/s" % rate(lambda: x.zero_(), 4 * n))
print("read only (sum):      %.2f TB/s" % rate(lambda: x.sum(), 4 * n))
print("copy (read + write):  %.2f TB/s" % rate(lambda: y.copy_(x), 8 * n))
