#!/usr/bin/env python3
"""Drives tools/probe/libwave1_probe.so: cycles per 32-pixel step of a one-wave-per-SIMD weight-gradient K loop (compute side)."""
import ctypes as C
import os
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class W1Args(C.Structure):
    _fields_ = [('out', C.c_void_p), ('sink', C.c_void_p), ('iters', C.c_int), ('barrier', C.c_int)]


lib = C.CDLL(os.path.join(HERE, 'libwave1_probe.so'))
lib.wave1_probe.argtypes = [C.POINTER(W1Args), C.c_int, C.c_int, C.c_void_p]
out = torch.zeros(4096, dtype=torch.int64, device='cuda')
sink = torch.zeros(256 * 256 * 4, dtype=torch.float32, device='cuda')
stream = torch.cuda.current_stream().cuda_stream
for mode, nm, nmf in ((0, '128 x 128 per wave: 64 MFMAs + 32 reads per step', 64), (1, '128 x 144 per wave: 72 MFMAs + 34 reads per step', 72)):
    for barrier in (0, 1):
        a = W1Args(out=out.data_ptr(), sink=sink.data_ptr(), iters=2000, barrier=barrier)
        for _ in range(3):
            assert lib.wave1_probe(C.byref(a), mode, 256, stream) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.wave1_probe(C.byref(a), mode, 256, stream)
        e1.record()
        torch.cuda.synchronize()
        cyc = out[:256].float().mean().item() / a.iters
        ms = e0.elapsed_time(e1)
        tf = 256 * 4 * a.iters * nmf * 2 * 16 * 16 * 32 / ms / 1e9
        print('%-56s barrier %d: %7.1f cycles per step (MFMA alone %d x 16 = %d)  %7.1f TFLOP/s  clock %.2f GHz'
              % (nm, barrier, cyc, nmf, nmf * 16, tf, out[:256].float().mean().item() / ms / 1e6), flush=True)
