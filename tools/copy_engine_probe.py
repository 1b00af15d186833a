"""Which engine moves a page-locked H2D / D2H copy on this runtime, and how fast, with kernels running beside it?
Run under different environments (HSA_ENABLE_SDMA, GPU_FORCE_BLIT_COPY_SIZE, ...); prints GB/s alone and the slowdown of a
concurrent compute stream.     python tools/copy_engine_probe.py"""
import os, sys, time
import torch
n = 40 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
a = torch.randn(4096, 4096, device="cuda")
cs, ks = torch.cuda.Stream(), torch.cuda.Stream()
def copies(k):
    with torch.cuda.stream(cs):
        for _ in range(k):
            h.copy_(d, non_blocking=True)
def gemms(k):
    with torch.cuda.stream(ks):
        for _ in range(k):
            a @ a
for _ in range(2):
    copies(3); gemms(3); torch.cuda.synchronize()
t0 = time.perf_counter(); copies(20); torch.cuda.synchronize(); tc = time.perf_counter() - t0
t0 = time.perf_counter(); gemms(40); torch.cuda.synchronize(); tg = time.perf_counter() - t0
t0 = time.perf_counter(); copies(20); gemms(40); torch.cuda.synchronize(); tb = time.perf_counter() - t0
print("env %s | D2H alone %.1f GB/s | 40 GEMMs alone %.1f ms | both together %.1f ms (sum %.1f)" %
      ({k: os.environ[k] for k in ("HSA_ENABLE_SDMA", "GPU_FORCE_BLIT_COPY_SIZE", "GPU_BLIT_ENGINE_TYPE") if k in os.environ},
       20 * n / tc / 1e9, tg * 1e3, tb * 1e3, (tc + tg) * 1e3))
