#!/usr/bin/env python3
"""Does a tensor written by kernel A come back from the 256 MiB Infinity Cache when kernel B reads it next?  (the
BatchNorm passes of the 38 x 38 / 19 x 19 stages - 24 .. 95 MB tensors just written by the previous kernel - run at 4.1 - 4.7 TB/s,
slower than the 300 MB ones.)  Not part of the library: plain torch kernels on one stream, HIP events around B only.

For sizes 12 MB .. 1.5 GB (fp16 elements):
  A  = y.copy_(x)                 writes y (and reads x, same size)
  B1 = (y > 0).sum-free read:     torch.amax(y)            read-only consumer
  B2 = z = y * 2 (out=z)          read + write consumer (what bn_act_fwd / bn_act_bwd_apply are)
timed (i) straight after A ("warm"), (ii) after a 2 GB flush kernel between A and B ("cold").  If the Infinity Cache serves a
producer -> consumer pair, warm is faster than cold up to about its capacity and equal beyond.
"""
import torch

dev = 'cuda'
flush_src = torch.empty(1 << 30, dtype=torch.float16, device=dev)      # 2 GB
flush_dst = torch.empty_like(flush_src)


def timed(fn, setup, reps=12):
    best = []
    for _ in range(reps):
        setup()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best.append(a.elapsed_time(b))
    best.sort()
    return best[len(best) // 2]


print('%8s | %-30s | %-30s' % ('MB', 'read-only consumer  warm / cold', 'read + write consumer  warm / cold'))
for mb in (12, 24, 48, 72, 96, 144, 192, 256, 384, 768, 1536):
    n = mb * (1 << 20) // 2
    x = torch.randn(n, dtype=torch.float16, device=dev)
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    out = torch.empty((), dtype=torch.float16, device=dev)

    def produce():
        y.copy_(x)

    def produce_flush():
        y.copy_(x)
        flush_dst.copy_(flush_src)

    r_w = timed(lambda: torch.amax(y, dim=0, out=out), produce)
    r_c = timed(lambda: torch.amax(y, dim=0, out=out), produce_flush)
    w_w = timed(lambda: torch.mul(y, 2, out=z), produce)
    w_c = timed(lambda: torch.mul(y, 2, out=z), produce_flush)
    gb = mb * (1 << 20) / 1e9
    print('%8d | %7.4f ms %6.0f GB/s / %7.4f ms %6.0f GB/s | %7.4f ms %6.0f GB/s / %7.4f ms %6.0f GB/s'
          % (mb, r_w, gb / r_w * 1e3, r_c, gb / r_c * 1e3, w_w, 2 * gb / w_w * 1e3, w_c, 2 * gb / w_c * 1e3), flush=True)
    del x, y, z
