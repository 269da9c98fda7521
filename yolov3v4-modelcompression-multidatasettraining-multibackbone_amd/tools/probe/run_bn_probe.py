#!/usr/bin/env python3
"""The library's BatchNorm + activation pass (yh_bn_act_fwd) and the backward pair (yh_bn_act_bwd_reduce / _apply) on the tensor
sizes of YOLOv3-608 batch 64, timed with the operand (i) just written by the preceding kernel ("warm": it may still sit in the
256 MiB Infinity Cache, tools/probe/run_mall_probe.py) and (ii) after a 2 GB flush ("cold"), next to a plain torch kernel with the
same traffic.  Tells whether the small passes of the training step (19 x 19 / 38 x 38: 3.0 - 4.7 TB/s in
profiles/r05_train_layers_final.txt) are slow because their operand is cold or because of their own launch geometry."""
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from engine import hiplib

lib = hiplib.load()
dev = 'cuda'
S = hiplib.stream_ptr()
P = hiplib.ptr
flush_src = torch.empty(1 << 30, dtype=torch.float16, device=dev)
flush_dst = torch.empty_like(flush_src)
ws = torch.empty(1 << 24, dtype=torch.float32, device=dev)


def timed(fn, setup, reps=10):
    t = []
    for _ in range(reps):
        setup()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        t.append(a.elapsed_time(b))
    t.sort()
    return t[len(t) // 2]


print('%-22s %8s | %-27s | %-27s | %-27s | %s' % ('tensor', 'MB', 'bn_act_fwd warm / cold (TB/s)', 'bwd reduce warm / cold', 'bwd apply warm / cold', 'torch z*2 warm / cold'))
for hw, c in ((19, 512), (19, 1024), (38, 256), (38, 512), (76, 128), (76, 256), (152, 64), (152, 128), (304, 64)):
    px = 64 * hw * hw
    z = torch.randn(px, c, dtype=torch.float16, device=dev)
    src = torch.randn(px, c, dtype=torch.float16, device=dev)
    skew = int(os.environ.get('YH_PROBE_SKEW', '0')) // 2       # elements: dy starts this many bytes into its allocation (HBM channel / bank alignment against z)
    dy_store = torch.randn(px * c + skew, dtype=torch.float16, device=dev)
    dy = dy_store[skew:].view(px, c)
    y = torch.empty_like(z)
    par = [torch.rand(c, dtype=torch.float32, device=dev) + 0.5 for _ in range(4)]
    sums = [torch.zeros(c, dtype=torch.float32, device=dev) for _ in range(2)]
    d = hiplib.BnDesc(z=P(z), dy=P(dy), out=P(y), gamma=P(par[0]), beta=P(par[1]), mean=P(par[2]), invstd=P(par[3]), sum=P(sums[0]),
                      sumsq=P(sums[1]), pixels=px, n=64, h=hw, w_in=hw, c=c, ldz=c, lddy=c, ldr=0, ldo=c, act=1, ups=1, dtype=hiplib.YH_F16,
                      slope=0.1, eps=1e-5, momentum=0.1, nparts=0, ws=P(ws), ws_floats=ws.numel())

    def produce():
        z.copy_(src)
        dy.copy_(src)

    def produce_flush():
        produce()
        flush_dst.copy_(flush_src)

    res = []
    for fn, nbytes in ((lambda: hiplib.check(lib.yh_bn_act_fwd(C.byref(d), S), 'fwd'), 2),
                       (lambda: hiplib.check(lib.yh_bn_act_bwd_reduce(C.byref(d), S), 'red'), 2),
                       (lambda: hiplib.check(lib.yh_bn_act_bwd_apply(C.byref(d), S), 'app'), 3),
                       (lambda: torch.mul(z, 2, out=y), 2)):
        w, cc = timed(fn, produce), timed(fn, produce_flush)
        gb = nbytes * px * c * 2 / 1e9
        res.append('%7.4f %5.2f / %7.4f %5.2f' % (w, gb / w, cc, gb / cc))
    print('%-22s %8.1f | %s' % ('%dx%d x %d' % (hw, hw, c), px * c * 2 / 1e6, ' | '.join(res)), flush=True)
    del z, src, dy, dy_store, y
