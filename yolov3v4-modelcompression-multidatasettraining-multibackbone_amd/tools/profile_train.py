#!/usr/bin/env python3
"""Per-op GPU time of one HIP training step (forward plan + backward plan), from the plan executor's HIP events.

    python tools/profile_train.py --batch 64 --size 608 [--precision fp16] [--top 40]
"""
import argparse
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default=os.path.join(os.path.dirname(HERE), 'cfg', 'yolov3', 'yolov3.cfg'))
    ap.add_argument('--size', type=int, default=608)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--precision', default='fp16')
    ap.add_argument('--top', type=int, default=0, help='only the N slowest ops (0 = all, in execution order)')
    args = ap.parse_args()
    from models import Darknet
    from engine.train import TrainEngine
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = Darknet(args.cfg, (args.size, args.size)).to(dev).train()
    eng = TrainEngine(model, args.precision)
    x = torch.rand(args.batch, 3, args.size, args.size, device=dev)
    heads = eng.forward(x)
    eng.backward([torch.randn_like(h) * 1e-3 for h in heads])
    torch.cuda.synchronize()
    plan, lib = eng._current, eng.lib
    rows = []
    for key, log in (('fwd', plan['fwd_ops']), ('bwd', plan['bwd_ops'])):
        handle = plan[key]
        lib.yh_plan_set_timing(handle, 1)
        acc = None
        reps = 3
        for _ in range(reps):
            if key == 'fwd':
                eng.forward(x)
            else:
                eng.forward(x)
                eng.backward([torch.randn_like(h) * 1e-3 for h in heads])
            torch.cuda.synchronize()
            n = lib.yh_plan_num_ops(handle)
            buf = (C.c_float * n)()
            lib.yh_plan_get_timings(handle, buf, n)
            acc = list(buf) if acc is None else [a + b for a, b in zip(acc, buf)]
        lib.yh_plan_set_timing(handle, 0)
        es = 2 if args.precision == 'fp16' else 4
        for (what, d), ms in zip(log, acc):
            ms /= reps
            name = what.rstrip('0123456789')
            flops = byts = 0.0
            shape = ''
            if name in ('conv', 'dgrad', 'wgrad'):
                flops = 2.0 * d.n * d.ho * d.wo * d.cout * d.cin * d.kh * d.kw
                shape = '%dx%d %d->%d k%d s%d' % (d.h, d.w_in, d.cin, d.cout, d.kh, d.stride)
                byts = es * d.n * (d.h * d.w_in * d.cin + d.ho * d.wo * d.cout)
            elif hasattr(d, 'pixels') and hasattr(d, 'c'):
                shape = '%d px x %d' % (d.pixels, d.c)
                mult = {'bnstat': 1, 'bnact': 2, 'dbn': 2, 'dbnx': 3, 'dbias': 1, 'dres': 3, 'dadd': 3}.get(name, 2)
                byts = es * d.pixels * d.c * mult
            rows.append((key, what, shape, ms, flops / ms / 1e9 if ms > 0 else 0, byts / ms / 1e6 if ms > 0 else 0))
    total = sum(r[3] for r in rows)
    show = sorted(rows, key=lambda r: -r[3])[:args.top] if args.top else rows
    print('%-4s %-10s %-28s %9s %9s %8s' % ('plan', 'op', 'shape', 'ms', 'TFLOP/s', 'GB/s'))
    for key, what, shape, ms, tf, gbs in show:
        print('%-4s %-10s %-28s %9.4f %9.1f %8.0f' % (key, what, shape, ms, tf, gbs))
    groups = {}
    for key, what, shape, ms, tf, gbs in rows:
        g = groups.setdefault(what.rstrip('0123456789'), [0.0, 0])
        g[0] += ms
        g[1] += 1
    print()
    for name, (ms, n) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
        print('%-10s %4d %9.3f ms %5.1f%%' % (name, n, ms, 100 * ms / total))
    print('total %.3f ms GPU time per step (plans only; loss/optimizer excluded)' % total)


if __name__ == '__main__':
    main()
