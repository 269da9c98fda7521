#!/usr/bin/env python3
"""Per-op GPU time of one Darknet forward (HIP events recorded by the native plan executor).

    python tools/profile_layers.py --batch 32 --size 608 [--cfg cfg/yolov3/yolov3.cfg] [--precision fp16]

Prints one line per recorded op: kernel family/tile, shape, milliseconds, TFLOP/s (conv) and GB/s of
algorithmic activation traffic, plus totals per kernel family.
"""
import argparse
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.dirname(PKG))

import torch  # noqa: E402


def main():
    from bench import build_model, build_qmodel_synthetic, tile_name
    from engine import hiplib
    ap = argparse.ArgumentParser()
    ap.add_argument('--cfg', default=os.path.join(PKG, 'cfg', 'yolov3', 'yolov3.cfg'))
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--size', type=int, default=608)
    ap.add_argument('--precision', default='fp16')
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--tile', type=int, default=0, help='force one conv tile code for every layer (A/B runs)')
    args = ap.parse_args()
    if args.tile:
        os.environ['YOLO_HIP_TILE'] = str(args.tile)      # engine/plan.py reads it when the engine is created (generic ring tiles 21 - 35; 3x3: 41 - 43)
    dev = torch.device('cuda', 0)
    model = build_qmodel_synthetic(args.cfg, args.size, dev) if args.precision == 'int8' else \
        build_model(args.cfg, args.size, args.precision, dev)
    x = torch.rand(args.batch, 3, args.size, args.size, device=dev)
    with torch.no_grad():
        model(x)
    eng = model.__dict__['_hip_engine']
    plan = eng._plans[tuple(x.shape)]
    lib, handle = eng.lib, plan['handle']
    n = lib.yh_plan_num_ops(handle)
    hiplib.check(lib.yh_plan_set_timing(handle, 1), 'timing')
    buf = (C.c_float * n)()
    tot = [0.0] * n
    for it in range(args.iters + 1):
        with torch.no_grad():
            model(x)
        torch.cuda.synchronize()
        hiplib.check(lib.yh_plan_get_timings(handle, buf, n), 'get')
        if it:
            for i in range(n):
                tot[i] += buf[i] / args.iters
    esz = {'fp16': 2, 'fp32': 4, 'int8': 1}[args.precision]
    vals = {('conv%d' % v.block if v.src.kind != 'input' else 'stem%d' % v.block): v for v in plan['values'] if v.kind == 'conv'}
    fam = {}
    print('%-8s %-18s %-34s %9s %9s %9s' % ('op', 'kernel', 'shape', 'ms', 'TFLOP/s', 'GB/s'))
    for i, (what, d) in enumerate(plan['ops']):
        ms = tot[i]
        name, shape, fl, by = what.rstrip('0123456789'), '', 0.0, 0.0
        v = vals.get(what)
        if v is not None:
            fl = 2.0 * args.batch * v.Ho * v.Wo * v.C * v.k * v.k * v.src.C
            by = args.batch * (v.src.H * v.src.W * v.src.C + v.H * v.W * v.C * (2 if v.res is not None else 1)) * esz
            shape = '%dx%d %d->%d k%d s%d%s%s' % (v.src.H, v.src.W, v.src.C, v.C, v.k, v.stride, ' +res' if v.res is not None else '',
                                                 ' ups' if v.ups == 2 else '')
            if isinstance(d, hiplib.ConvDesc):
                name = 'igemm_' + tile_name(lib.yh_conv2d_tile(C.byref(d)))
        f = fam.setdefault(name, [0.0, 0.0, 0])
        f[0] += ms
        f[1] += fl
        f[2] += 1
        print('%-8s %-18s %-34s %9.4f %9.1f %9.0f' % (what, name, shape, ms, fl / ms / 1e9 if ms and fl else 0, by / ms / 1e6 if ms and by else 0))
    total = sum(tot)
    print('\n%-18s %6s %9s %7s %9s' % ('kernel', 'n', 'ms', '%', 'TFLOP/s'))
    for name, (ms, fl, cnt) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print('%-18s %6d %9.4f %6.1f%% %9.1f' % (name, cnt, ms, 100 * ms / total, fl / ms / 1e9 if fl else 0))
    allfl = sum(f[1] for f in fam.values())
    print('total %.4f ms / step, %.1f img/s (GPU time only), %.1f TFLOP/s' % (total, args.batch / total * 1e3, allfl / total / 1e9))


if __name__ == '__main__':
    main()
