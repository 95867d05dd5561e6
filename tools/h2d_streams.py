"""H2D bandwidth from pinned memory with 1, 2 and 4 copy streams (development aid for omr_host_batch: is one DMA stream
the limit of e2e_host_images_per_s?).  Usage: python tools/h2d_streams.py"""
import time

import torch

dev = torch.device("cuda:0")
N = 64 * 2480 * 3508  # one ring slot of the host batch: 64 A4 scans
src = [torch.empty(N, dtype=torch.uint8).pin_memory() for _ in range(4)]
dst = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(4)]
for ns in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(8):
            k = it % ns
            with torch.cuda.stream(streams[k]):
                dst[it % 4].copy_(src[it % 4], non_blocking=True)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    print("%d stream(s): %.1f GB/s" % (ns, 8 * N / t / 1e9), flush=True)
# smaller pieces on alternating streams (one scan per copy)
M = 2480 * 3508
for ns in (1, 2):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(256):
            with torch.cuda.stream(streams[it % ns]):
                dst[0][(it % 64) * M:(it % 64 + 1) * M].copy_(src[0][(it % 64) * M:(it % 64 + 1) * M], non_blocking=True)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    print("one scan per copy, %d stream(s): %.1f GB/s" % (ns, 256 * M / t / 1e9), flush=True)
