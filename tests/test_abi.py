"""The shared library must load without a GPU and export exactly the C ABI that include/yolo_hip.h declares."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from engine import hiplib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, 'include', 'yolo_hip.h')


def _declared():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(yh_[a-z0-9_]+)\s*\(', text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = hiplib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), 'libyolo_hip.so does not export ' + n
    assert sorted(hiplib.EXPORTS) == names, 'ctypes binding and header disagree'
    assert lib.yh_abi_version() == hiplib.ABI_VERSION
    assert b'misaligned' in lib.yh_error_string(-2)


def test_struct_layout_matches_c_compiler(tmp_path):
    """ctypes mirrors must have the C compiler's sizeof/offsetof (gcc, no GPU needed)."""
    structs = {'yh_conv_desc': hiplib.ConvDesc, 'yh_stem_desc': hiplib.StemDesc, 'yh_pool_desc': hiplib.PoolDesc,
               'yh_copy_desc': hiplib.CopyDesc, 'yh_add_desc': hiplib.AddDesc, 'yh_decode_desc': hiplib.DecodeDesc,
               'yh_dw_desc': hiplib.DwDesc, 'yh_se_desc': hiplib.SeDesc, 'yh_qcopy_desc': hiplib.QCopyDesc,
               'yh_qadd_desc': hiplib.QAddDesc, 'yh_letterbox_desc': hiplib.LetterboxDesc, 'yh_mosaic_desc': hiplib.MosaicDesc,
               'yh_stem_bwd_desc': hiplib.StemBwdDesc, 'yh_wgrad_desc': hiplib.WgradDesc, 'yh_bn_desc': hiplib.BnDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "yolo_hip.h"', 'int main(void){']
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines.append('return 0;}')
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-I', os.path.join(REPO, 'include'), str(src), '-o', str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got['%s.%s' % (cname, fname)]) == getattr(cls, fname).offset, '%s.%s' % (cname, fname)


def test_plan_api_runs_without_gpu():
    """Plan bookkeeping is host code: create/add/fixup/destroy work on a machine without a GPU."""
    lib = hiplib.load()
    h = lib.yh_plan_create()
    assert h
    d = hiplib.CopyDesc(n=1, h=2, w_in=2, c=8, ups=1, ldx=8, ldy=8, dtype=0)
    idx = lib.yh_plan_add(h, hiplib.OP_COPY, C.byref(d), C.sizeof(d))
    assert idx == 0 and lib.yh_plan_num_ops(h) == 1
    assert lib.yh_plan_add(h, hiplib.OP_COPY, C.byref(d), 4) == -1  # wrong size is rejected
    assert lib.yh_plan_add_fixup(h, 0, hiplib.CopyDesc.x.offset, 0, 0) == 0
    assert lib.yh_plan_add_fixup(h, 5, 0, 0, 0) == -5  # YH_ERANGE
    assert lib.yh_plan_run(h, None) == -1  # unbound slot -> YH_EINVAL, nothing launched
    lib.yh_plan_destroy(h)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(hiplib, '_lib', None)
    monkeypatch.setattr(hiplib, 'LIB_PATH', '/nonexistent/libyolo_hip.so')
    with pytest.raises(hiplib.HipLibraryError):
        hiplib.load()


def _conv(dtype, n, hw, cin, cout, k, s, res=False, stats=False, act=1):
    esz = 1 if dtype == hiplib.YH_I8 else 2
    step = 64 if dtype == hiplib.YH_I8 else 32
    ho = (hw + 2 * (k // 2) - k) // s + 1
    d = hiplib.ConvDesc(n=n, h=hw, w_in=hw, cin=cin, ho=ho, wo=ho, cout=cout, kh=k, kw=k, stride=s, pad=k // 2, ldx=cin, ldr=cout if res else 0,
                        ldy=cout, cin_k=-(-cin // step) * step, m_pad=-(-cout // 128) * 128, act=act, slope=0.1, ups=1, dtype=dtype,
                        acc_scale=1.0, out_scale=1.0)
    d.x = d.w = d.bias = d.y = 4096          # any aligned non-null address: the picker reads alignment only
    if res:
        d.res = 4096
        d.q_rx = d.q_ra = d.q_scale_x = d.q_scale_a = d.q_inv_scale_sum = 1.0
    if stats:
        d.stats_ws, d.stats_ws_floats = 4096, 1 << 40
    return d


def test_kernel_selection_on_the_headline_network():
    """Which kernel family a layer of YOLOv3-608 at batch 64 runs on (yh_conv2d_tile is host code): the few-channel layers of the
    608 / 304 stages on the streaming kernels, the 3x3 layers from 128 channels on the halo ping-pong kernel, 1x1 layers on the ring."""
    lib = hiplib.load()
    F16, I8 = hiplib.YH_F16, hiplib.YH_I8
    tile = lambda d: lib.yh_conv2d_tile(C.byref(d))
    # (dtype, hw, cin, cout, k, s, residual) -> tile code; inference forms
    expect = [((F16, 608, 32, 64, 3, 2, False), 72), ((F16, 304, 32, 64, 3, 1, True), 72), ((I8, 608, 32, 64, 3, 2, False), 72),
              ((I8, 304, 32, 64, 3, 1, True), 72), ((I8, 304, 64, 128, 3, 2, False), 72), ((I8, 152, 64, 128, 3, 1, True), 72),
              ((F16, 304, 64, 32, 1, 1, False), 71), ((I8, 152, 128, 64, 1, 1, False), 71),
              ((F16, 76, 128, 256, 3, 1, True), 43), ((F16, 38, 256, 512, 3, 1, True), 43), ((F16, 19, 512, 1024, 3, 1, True), 43),
              ((I8, 76, 128, 256, 3, 1, False), 43), ((I8, 38, 256, 512, 3, 1, True), 43), ((F16, 152, 64, 128, 3, 1, True), 43),
              # round 6 (profiles/r06_ring_tile_sweep.txt): int8 with two channel chunks AND the fused quantised shortcut is ahead on the ring
              ((I8, 76, 128, 256, 3, 1, True), 26)]
    for (dt, hw, cin, cout, k, s, res), want in expect:
        assert tile(_conv(dt, 64, hw, cin, cout, k, s, res)) == want, (dt, hw, cin, cout, k, s, res)
    # 1x1 layers of the deeper stages and the fp16 64 -> 128 stride-2 layer stay on the LDS-DMA ring / ping-pong kernels
    for d in (_conv(F16, 64, 76, 256, 128, 1, 1), _conv(F16, 64, 38, 512, 256, 1, 1), _conv(F16, 64, 304, 64, 128, 3, 2)):
        assert tile(d) not in (43, 71, 72)
    # the training forward (statistics epilogue): the streaming 3x3 kernel carries them in fp16, one partial row per wave of its launch
    d = _conv(F16, 64, 304, 32, 64, 3, 1, stats=True, act=0)
    assert tile(d) == 72 and lib.yh_conv2d_stats_rows(C.byref(d)) == 512 * 4
    d = _conv(F16, 2, 40, 32, 64, 3, 1, stats=True, act=0)        # a small grid keeps the ring kernel: rows per (128-pixel tile, wave column)
    assert tile(d) not in (72,) and lib.yh_conv2d_stats_rows(C.byref(d)) == -(-2 * 40 * 40 // 128) * 2
    # what the streaming kernels do not do is refused when asked for explicitly, not approximated
    bad = _conv(F16, 64, 304, 32, 48, 3, 1)
    bad.tile = 72
    assert lib.yh_conv2d_fwd(C.byref(bad), None) != 0
