#!/usr/bin/env python3
"""Drives tools/probe/liblds_probe.so: LDS fragment-read rate per CU (bytes per clock), alone and with 24 MFMAs per 20 reads."""
import ctypes as C
import os
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class LdsArgs(C.Structure):
    _fields_ = [('out', C.c_void_p), ('iters', C.c_int), ('mfma', C.c_int), ('reads', C.c_int)]


lib = C.CDLL(os.path.join(HERE, 'liblds_probe.so'))
lib.lds_probe.argtypes = [C.POINTER(LdsArgs), C.c_int, C.c_int, C.c_int, C.c_void_p]
out = torch.zeros(8192, dtype=torch.int64, device='cuda')
stream = torch.cuda.current_stream().cuda_stream
names = {0: 'ds_read_b64_tr_b16', 1: 'ds_read_b64', 2: 'ds_read_b128'}
for kind in (0,):
    for waves in (4, 8, 12):
        for mfma in (1, 2):
            for reads in (10,):
                a = LdsArgs(out=out.data_ptr(), iters=2000, mfma=mfma, reads=reads)
                for _ in range(2):
                    assert lib.lds_probe(C.byref(a), kind, 256, waves, stream) == 0
                torch.cuda.synchronize()
                cyc = out[:256].float().mean().item() / a.iters
                nbytes = waves * reads * 2 * 512
                print('%-36s waves %2d  reads/iter %2d  mfma %d:  %7.1f cycles per iteration  %6.1f B/clk/CU%s'
                      % (names[kind] + (' rolling refresh' if mfma == 2 else ''), waves, reads * 2, mfma, cyc, nbytes / cyc, ('   (MFMA alone: %d x 24 x 16 / 4 SIMDs = %d)' % (waves, waves * 24 * 16 // 4)) if mfma else ''), flush=True)
