#!/usr/bin/env python3
"""Drives tools/probe/libdma_probe.so: service rate of the global -> LDS (LDS-DMA) and global -> register paths per CU for the piece
shapes the convolution kernels use.  Prints bytes / clock / CU (s_memtime cycles of the workgroups) and aggregate TB/s (HIP events)."""
import ctypes as C
import os
import sys
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class ProbeArgs(C.Structure):
    _fields_ = [('src', C.c_void_p), ('out', C.c_void_p), ('wg_stride', C.c_long), ('wave_stride', C.c_long), ('iter_stride', C.c_long),
                ('row_bytes', C.c_int), ('row_stride', C.c_int), ('iters', C.c_int), ('depth', C.c_int), ('swz', C.c_int), ('dz_waves', C.c_int),
                ('src2', C.c_void_p), ('wg_stride2', C.c_long), ('wave_stride2', C.c_long), ('iter_stride2', C.c_long),
                ('row_bytes2', C.c_int), ('row_stride2', C.c_int)]


def main():
    lib = C.CDLL(os.path.join(HERE, 'libdma_probe.so'))
    lib.dma_probe.argtypes = [C.POINTER(ProbeArgs), C.c_int, C.c_int, C.c_int, C.c_void_p]
    buf = torch.randint(0, 255, (3 << 30,), dtype=torch.uint8, device='cuda')
    buf2 = torch.randint(0, 255, (2 << 30,), dtype=torch.uint8, device='cuda')
    out = torch.zeros(4096, dtype=torch.int64, device='cuda')
    stream = torch.cuda.current_stream().cuda_stream
    iters = 512

    def run(name, mode, grid, waves, depth, rb, rs, swz=0, second=None, iters=iters, same=False):
        rows = 1024 // rb
        a = ProbeArgs(src=buf.data_ptr(), out=out.data_ptr(), row_bytes=rb, row_stride=rs, iters=iters, depth=depth, swz=swz, dz_waves=waves)
        nw1 = waves if second is None else second[0]
        a.dz_waves = nw1
        a.wave_stride = rows * rs
        a.iter_stride = nw1 * rows * rs
        a.wg_stride = 0 if same else iters * a.iter_stride
        total = grid * nw1 * iters * 1024
        if second is not None:
            _, rb2, rs2 = second
            rows2 = 1024 // rb2
            nw2 = waves - nw1
            a.src2 = buf2.data_ptr()
            a.row_bytes2, a.row_stride2 = rb2, rs2
            a.wave_stride2 = rows2 * rs2
            a.iter_stride2 = nw2 * rows2 * rs2
            a.wg_stride2 = 0 if same else iters * a.iter_stride2
            total += grid * nw2 * iters * 1024
            assert grid * iters * a.iter_stride2 <= buf2.numel()
        assert (1 if same else grid) * iters * a.iter_stride + 4096 <= buf.numel(), name
        for _ in range(2):
            rc = lib.dma_probe(C.byref(a), mode, grid, waves, stream)
            assert rc == 0, rc
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.dma_probe(C.byref(a), mode, grid, waves, stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        cyc = out[:grid].float().mean().item()
        print('%-64s grid %3d waves %2d depth %2d  %6.2f B/clk/CU  %6.2f TB/s  (%.3f ms, %.0f cycles)'
              % (name, grid, waves, depth, total / grid / cyc, total / ms / 1e9, ms, cyc), flush=True)

    for mode, nm in ((0, 'free-running'), (3, 'one barrier per step, all waves issue together'), (2, 'three barriers per step, groups take turns')):
        for depth in (4, 9):
            run('dz + x streams, %s' % nm, mode, 256, 12, depth, 256, 512, swz=1, second=(8, 128, 256))
    run('dz + x streams, groups take turns, 128 workgroups', 2, 128, 12, 9, 256, 512, swz=1, second=(8, 128, 256))
    run('dz + x streams, groups take turns, same streams (L2 hits)', 2, 256, 12, 9, 256, 512, swz=1, second=(8, 128, 256), same=True)
    run('dz + x streams, one barrier, same streams (L2 hits)', 3, 256, 12, 9, 256, 512, swz=1, second=(8, 128, 256), same=True)


if __name__ == '__main__':
    main()
