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
               'yh_qadd_desc': hiplib.QAddDesc, 'yh_letterbox_desc': hiplib.LetterboxDesc, 'yh_mosaic_desc': hiplib.MosaicDesc}
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
