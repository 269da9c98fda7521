#!/usr/bin/env python3
"""A/B of the weight-gradient kernels on the layer shapes of a training step (one process, same box): for every shape, each mode of
YH_WGRAD_HALO (1 = conv_wgrad_halo_kernel, 2 = conv_wgrad_roll_kernel) is checked against mode 0 (im2col kernels) on random operands
and timed with HIP events over `--reps` back-to-back launches (kernel + its reduce launch).

    python tools/wgrad_ab.py [--batch 64] [--reps 20] [--shapes yolov3] [--env K=V ...variants]
"""
import argparse
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
sys.path[:0] = [PKG, os.path.join(os.path.dirname(PKG), 'tests')]
import torch  # noqa: E402
from engine import hiplib  # noqa: E402
import ops_harness as oh  # noqa: E402

SHAPES = {
    'yolov3': [(76, 76, 128, 256, 3), (38, 38, 256, 512, 3), (19, 19, 512, 1024, 3), (152, 152, 64, 128, 3)],
    'k76': [(76, 76, 128, 256, 3)], 'k38': [(38, 38, 256, 512, 3)], 'k19': [(19, 19, 512, 1024, 3)],
    'yolov3_1x1': [(76, 76, 256, 128, 1), (38, 38, 512, 256, 1), (19, 19, 1024, 512, 1), (152, 152, 128, 64, 1), (304, 304, 64, 32, 1)],
}


def set_env(sets):
    for kk in [e for e in os.environ if e.startswith('YH_WGRAD')]:
        del os.environ[kk]
    os.environ.update(sets)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--reps', type=int, default=40)
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--shapes', default='yolov3')
    ap.add_argument('--variants', nargs='*', default=['YH_WGRAD_HALO=1', 'YH_WGRAD_HALO=2'],
                    help='environment settings to compare, comma-separated K=V lists, e.g. YH_WGRAD_HALO=2,YH_WGRAD_ROLL_STAGES=6')
    ap.add_argument('--no-check', action='store_true')
    args = ap.parse_args()
    lib = hiplib.load()
    dev = 'cuda'
    g = torch.Generator().manual_seed(0)
    print('%-28s %-52s %9s %9s %9s %10s' % ('shape', 'variant', 'median ms', 'min ms', 'TFLOP/s', 'max rel err'))
    for (H, W, cin, cout, k) in SHAPES[args.shapes]:
        N = args.batch
        x = (torch.randn(N, H, W, cin, generator=g) * 0.7).half().to(dev)
        dz = (torch.randn(N, H, W, cout, generator=g) * 0.05).half().to(dev)
        flops = 2.0 * N * H * W * cin * cout * k * k
        ref = None
        if not args.no_check:
            os.environ['YH_WGRAD_HALO'] = '0'
            ref = oh.wgrad(lib, hiplib.YH_F16, x, dz, cin, cout, k, 1, (k - 1) // 2)
            torch.cuda.synchronize()
        setups = []
        for var in args.variants:
            sets = dict(kv.split('=') for kv in var.split(','))
            set_env(sets)
            stamps = int(sets.get('YH_WGRAD_ROLL_ABL', '0')) & 8 != 0      # s_memtime sums of one workgroup behind the partial tiles
            dw = torch.zeros((cout, cin, k, k), device=dev, dtype=torch.float32)
            d = oh.WgradDesc(x=oh.P(x), dz=oh.P(dz), dw=oh.P(dw), n=N, h=H, w_in=W, cin=cin, ho=H, wo=W, cout=cout, kh=k, kw=k,
                             stride=1, pad=(k - 1) // 2, ldx=cin, lddz=cout, dtype=hiplib.YH_F16, splits=0)
            need = int(lib.yh_conv2d_wgrad_workspace(C.byref(d)))
            ws = torch.zeros((max(need, 1) + 64,), device=dev, dtype=torch.float32)
            d.ws, d.ws_floats = oh.P(ws), need + (64 if stamps else 0)
            code = int(lib.yh_conv2d_wgrad_kernel(C.byref(d)))
            oh.call(lib, 'yh_conv2d_wgrad', d)
            torch.cuda.synchronize()
            err = float('nan')
            if ref is not None and int(sets.get('YH_WGRAD_ROLL_ABL', '0')) & 7 == 0:
                err = ((dw - ref).abs().max() / ref.abs().max()).item()
            setups.append(dict(var=var, sets=sets, d=d, dw=dw, ws=ws, need=need, code=code, err=err, stamps=stamps, ms=[]))
        # interleaved rounds: the clock / thermal state drifts over a run by more than the differences looked for (the same kernel
        # measured 0.26 and 0.23 ms at the start and the end of one process), so every variant is timed once per round
        for rnd in range(args.rounds):
            for su in setups:
                set_env(su['sets'])
                for _ in range(3):
                    oh.call(lib, 'yh_conv2d_wgrad', su['d'])
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    oh.call(lib, 'yh_conv2d_wgrad', su['d'])
                e1.record()
                torch.cuda.synchronize()
                su['ms'].append(e0.elapsed_time(e1) / args.reps)
        for su in setups:
            if su['stamps']:
                t = su['ws'][su['need']:su['need'] + 48].view(torch.int64).cpu().view(3, 8)
                names = ('reads', 'barrier1', 'mfma', 'barrier2', 'wait+roll', 'barrier3', 'issue')
                for gq in range(3):
                    ns = max(int(t[gq, 7]), 1)
                    print('    group %d (%d steps), cycles per step: %s  total %d' % (gq, ns, '  '.join('%s %d' % (nm, int(t[gq, kk]) // ns) for kk, nm in enumerate(names)),
                                                                                 sum(int(t[gq, kk]) for kk in range(7)) // ns))
            ms = sorted(su['ms'])
            med = ms[len(ms) // 2]
            print('%-28s %-52s %9.4f %9.4f %9.1f %10.2e' % ('%dx%d %d->%d k%d b%d' % (H, W, cin, cout, k, N), '%s [kernel %d]' % (su['var'], su['code']),
                                                           med, ms[0], flops / med / 1e9, su['err']), flush=True)
        del x, dz
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
