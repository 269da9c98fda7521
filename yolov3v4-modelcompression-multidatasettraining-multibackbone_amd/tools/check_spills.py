#!/usr/bin/env python3
"""`make check-spills`: no shipped kernel instantiation may use scratch memory.

DESIGN.md 3 ("Toolchain facts"): a register spill is a memory operation on the same in-order `vmcnt` queue as the LDS-DMA ring, so
one spilled value inside a K loop drains the ring every iteration (measured 2 - 3x).  This script compiles every HIP source to
assembly for gfx950 (device side only), reads the amdhsa kernel metadata and fails when a kernel has
`.private_segment_fixed_size > 0` or spilled registers.  ALLOW lists kernels where scratch is not on a hot path, with the reason.

    python tools/check_spills.py [sources...]        # default: every csrc/*.hip of the Makefile's SRCS
"""
import concurrent.futures as cf
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, '..', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FILT = os.environ.get('CXXFILT', 'c++filt')
ALLOW = {
    # kernel-name regex -> reason
    r'conv_igemm_glds_kernelIDF16_(f|DF16_)Li(256ELi128|128ELi256)ELi\d+ELi\d+ELi3ELi0E':   # mangled: c++filt here predates _Float16
        'one 8-byte value parked across the GENERIC store form only (2x-nearest / stride-2 phase-scatter epilogues, run-time '
        'activation); the K loop and the plain dense epilogue of these instantiations are spill-free (asm: the scratch pair sits '
        'between the ups >= 2 column set-up and its stores)',
    r'mosaic_affine_hsv(_cv2)?_kernel':
        'a run-time indexed per-thread table (the four mosaic parts), not a register spill (vgpr spills 0); gather-latency-bound '
        'kernel without an LDS-DMA ring',
}


def kernels_of(src):
    flags = ['-O3', '-std=c++17', '--offload-arch=gfx950', '--cuda-device-only', '-S', '-o', '-']
    if os.path.basename(src) in ('nms.hip', 'quant.hip', 'loss.hip', 'calib.hip'):
        flags.insert(0, '-ffp-contract=off')
    asm = subprocess.run([HIPCC] + flags + [src], check=True, capture_output=True, text=True, cwd=CSRC).stdout
    meta = asm[asm.index('amdhsa.kernels:'):] if 'amdhsa.kernels:' in asm else ''
    out = []
    for block in re.split(r'\n  - ', meta)[1:]:
        def field(name, default='0'):
            m = re.search(r'\.%s:\s*(\S+)' % name, block)
            return m.group(1) if m else default
        out.append(dict(name=field('name', '?'), scratch=int(field('private_segment_fixed_size')), vgpr=int(field('vgpr_count')),
                        agpr=int(field('agpr_count')), sgpr_spill=int(field('sgpr_spill_count')), vgpr_spill=int(field('vgpr_spill_count')),
                        lds=int(field('group_segment_fixed_size'))))
    return out


def demangle(names):
    try:
        res = subprocess.run([FILT], input='\n'.join(names), capture_output=True, text=True, check=True).stdout.split('\n')
        return dict(zip(names, res))
    except Exception:
        return {n: n for n in names}


def main():
    srcs = sys.argv[1:]
    if not srcs:
        mk = open(os.path.join(CSRC, 'Makefile')).read()
        srcs = re.search(r'^SRCS := (.*)$', mk, re.M).group(1).split()
    srcs = [s if os.path.isabs(s) else os.path.join(CSRC, s) for s in srcs]
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        per_src = list(ex.map(kernels_of, srcs))
    bad, total, capped = [], 0, 0
    for src, ks in zip(srcs, per_src):
        names = demangle([k['name'] for k in ks])
        for k in ks:
            total += 1
            capped += k['vgpr'] + k['agpr'] >= 249
            pretty = names[k['name']]
            if k['scratch'] or k['vgpr_spill']:
                why = next((r for pat, r in ALLOW.items() if re.search(pat, pretty) or re.search(pat, k['name'])), None)
                tag = 'allowed: %s' % why.split(';')[0][:60] if why else 'SPILL'
                print('%s: %s  scratch %d B  vgpr spills %d  sgpr spills %d  (%d vgpr + %d agpr)  [%s]'
                      % (os.path.basename(src), pretty[:150], k['scratch'], k['vgpr_spill'], k['sgpr_spill'], k['vgpr'], k['agpr'], tag))
                if not why:
                    bad.append(pretty)
    print('check-spills: %d kernels in %d sources, %d at the 249+ register cap, %d with scratch' % (total, len(srcs), capped, len(bad)))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
