"""`make check-spills` as a CPU-tier test: the kernels that carry an LDS-DMA ring must not use scratch memory -
a register spill is a memory operation on the ring's in-order `vmcnt` queue and drains it every iteration ("Toolchain
facts").  tools/check_spills.py compiles the sources to gfx950 assembly (device side only; hipcc cross-compiles without a GPU) and
reads `.private_segment_fixed_size` / the spill counts of every kernel instantiation.  The gate rotted once within six hours of being
introduced because nothing ran it (round 4: conv3x3_hpp_kernel<int8, 7, 0>, 11 SGPR spills after an epilogue change); here it runs with
the test tier on the sources whose kernels sit at their register caps (conv_igemm_k64.hip alone compiles for six minutes: it joins with
YOLO_TEST_SPILLS_ALL=1, and `make check-spills` in csrc/ covers every source)."""
import os
import subprocess
import sys

import conftest

SOURCES = ['conv_halo_pp.hip', 'conv_wgrad.hip', 'conv_wgrad_roll.hip', 'conv_stem_mfma.hip'] + (['conv_igemm_k64.hip'] if os.environ.get('YOLO_TEST_SPILLS_ALL') else [])


def test_no_kernel_of_the_dma_ring_sources_uses_scratch():
    tool = os.path.join(conftest.PKG, 'tools', 'check_spills.py')
    res = subprocess.run([sys.executable, tool] + SOURCES, capture_output=True, text=True, timeout=1500)
    tail = (res.stdout + res.stderr)[-3000:]
    assert res.returncode == 0, tail
    assert 'check-spills:' in res.stdout and ' 0 with scratch' in res.stdout, tail
