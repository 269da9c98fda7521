"""COS-PTQ calibration on the device: what `utils/quantized/quantized_ptq_cos.py` runs when its tensors live
on a GPU.  Three services, all on libyolo_hip.so (a missing library raises - there is no ATen / MIOpen fallback):

* ``cos_search``  the 15-candidate (quantisers) / 8-candidate (shortcuts) cosine vote as ONE pass over the tensor
                  (reference quantized_ptq_cos.py:64-93, :838-912, :1153-1197) - `yh_ptq_cos_search`;
* ``absmax``      the range the quantised concat tracks (:1403-1449) - `yh_absmax`;
* ``conv2d``      the float-stream / fake-quantised-stream convolutions of calibration mode (:230-275, :288-296) through
                  `yh_conv2d_fwd` in fp32 (exact-product `v_mfma_f32_16x16x4_f32`, fp32 accumulation), NCHW in, NCHW view out;
                  depthwise blocks (groups == channels, the Mobilenet / GhostNet cfgs) through `yh_dwconv2d_fwd` in fp32.

The decision (an index) is read back by the caller: the modules keep their vote histograms as Python lists, exactly like the
reference, so one 4-byte device-to-host read per search is part of the contract.
"""
import ctypes as C

import torch

from . import hiplib

_lib_override = None      # tests inject the host emulation of the C ABI here


def _lib():
    return _lib_override if _lib_override is not None else hiplib.load()


def _dense(t):
    """``t`` as a dense fp32 block of memory in ANY element order (the sums do not care): no copy for contiguous tensors and for
    permuted views of contiguous tensors (the NHWC-backed outputs of ``conv2d`` below)."""
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    span = 1 + sum((n - 1) * s for n, s in zip(t.shape, t.stride()))
    if t.numel() == 0 or span != t.numel() or any(s < 0 for s in t.stride()):
        t = t.contiguous()
    return t


def cos_search(t, scale0, n, bits, clamp=True):
    """(best index, [cos_0 .. cos_{n-1}]) of the candidates scale0 * 2^j, j < n, on the int`bits` grid."""
    lib = _lib()
    t = _dense(t)
    lo, hi = float(-(1 << (bits - 1))), float((1 << (bits - 1)) - 1)
    with hiplib.on_device(t):
        need = int(lib.yh_ptq_search_workspace(t.numel()))
        ws = torch.empty(max(need, 8) // 8 + 1, dtype=torch.float64, device=t.device)
        out = torch.zeros(n + 1, dtype=torch.float64, device=t.device)      # [cos_0 .. cos_{n-1} | best (int32 bits)]
        best = out[n:].view(torch.int32)
        hiplib.check(lib.yh_ptq_cos_search(hiplib.ptr(t), t.numel(), float(scale0), int(n), lo, hi, 1 if clamp else 0, hiplib.ptr(ws),
                                           ws.numel() * 8, hiplib.ptr(out), hiplib.ptr(best), hiplib.stream_ptr()), 'ptq cos search')
        host = out.cpu()
    return int(host[n:].view(torch.int32)[0]), [float(v) for v in host[:n]]


def absmax(t):
    """max |t| as a 0-d device tensor."""
    lib = _lib()
    t = _dense(t)
    with hiplib.on_device(t):
        ws = torch.empty(1024, dtype=torch.float32, device=t.device)
        out = torch.zeros((), dtype=torch.float32, device=t.device)
        hiplib.check(lib.yh_absmax(hiplib.ptr(t), t.numel(), hiplib.ptr(ws), ws.numel() * 4, hiplib.ptr(out), hiplib.stream_ptr()), 'absmax')
    return out


def _round_up(v, m):
    return (v + m - 1) // m * m


def conv2d(x, w, b, stride, padding, dilation=(1, 1), groups=1):
    """fp32 NCHW convolution of calibration mode on the HIP conv kernels; returns an NCHW *view* of the NHWC result."""
    s, p, d = (v if isinstance(v, int) else v[0] for v in (stride, padding, dilation))
    sq = all(isinstance(v, int) or v[0] == v[-1] for v in (stride, padding, dilation))
    cout, cin, kh, kw = w.shape
    if groups != 1 and groups == x.shape[1] == cout and cin == 1 and d == 1 and sq and kh == kw:
        return _dwconv2d(x, w, b, s, p)      # the Mobilenet / GhostNet depthwise blocks (reference models.py:115-160 builds them with this class)
    if groups != 1 or d != 1 or not sq or kh != kw:
        raise NotImplementedError('device calibration lowers dense square and depthwise convolutions only (groups=%d, dilation=%s, '
                                  'kernel %dx%d): calibrate this graph on the CPU' % (groups, dilation, kh, kw))
    lib = _lib()
    F32 = hiplib.YH_F32
    x = x.detach().float().contiguous()
    w = w.detach().float().contiguous()
    n, _, h, wi = x.shape
    ho, wo = (h + 2 * p - kh) // s + 1, (wi + 2 * p - kw) // s + 1
    cin_p = _round_up(cin, 4)
    cin_k = _round_up(cin_p, 16)
    cout_p = _round_up(cout, 4)
    m_pad = _round_up(cout_p, 128)
    dev = x.device
    with hiplib.on_device(x):
        S = hiplib.stream_ptr()
        xh = torch.empty((n, h, wi, cin_p), dtype=torch.float32, device=dev)
        hiplib.check(lib.yh_nchw_to_nhwc(hiplib.ptr(x), hiplib.ptr(xh), n, cin, h, wi, cin_p, cin_p, F32, S), 'nchw_to_nhwc')
        packed = torch.empty(m_pad * kh * kw * cin_k, dtype=torch.float32, device=dev)
        bias = torch.empty(m_pad, dtype=torch.float32, device=dev)
        cb = torch.zeros(cout, dtype=torch.float32, device=dev) if b is None else b.detach().float().contiguous()
        hiplib.check(lib.yh_conv_pack_weights(F32, hiplib.ptr(w), hiplib.ptr(cb), None, None, None, None, 0.0, None, cout, cin, kh, kw,
                                              cin_k, m_pad, hiplib.ptr(packed), hiplib.ptr(bias), S), 'pack')
        y = torch.empty((n, ho, wo, cout_p), dtype=torch.float32, device=dev)
        desc = hiplib.ConvDesc(x=hiplib.ptr(xh), w=hiplib.ptr(packed), bias=hiplib.ptr(bias), res=None, y=hiplib.ptr(y), n=n, h=h, w_in=wi,
                               cin=cin_p, ho=ho, wo=wo, cout=cout_p, kh=kh, kw=kw, stride=s, pad=p, ldx=cin_p, ldr=0, ldy=cout_p,
                               cin_k=cin_k, m_pad=m_pad, act=0, slope=0.0,
                               ups=1, out_f32=1, dtype=F32, tile=0)
        hiplib.check(lib.yh_conv2d_fwd(C.byref(desc), S), 'calibration conv')
    return y[..., :cout].permute(0, 3, 1, 2)


def _dwconv2d(x, w, b, s, p):
    """Depthwise (groups == channels) fp32 convolution of calibration mode on ``yh_dwconv2d_fwd`` (csrc/depthwise.hip): NCHW in,
    NCHW view of the NHWC result out, like ``conv2d``."""
    lib = _lib()
    F32 = hiplib.YH_F32
    x = x.detach().float().contiguous()
    w = w.detach().float().contiguous()
    n, c, h, wi = x.shape
    k = w.shape[2]
    ho, wo = (h + 2 * p - k) // s + 1, (wi + 2 * p - k) // s + 1
    c_p = _round_up(c, 8)          # yh_dw_pack_weights packs 8-channel groups
    dev = x.device
    with hiplib.on_device(x):
        S = hiplib.stream_ptr()
        xh = torch.empty((n, h, wi, c_p), dtype=torch.float32, device=dev)
        hiplib.check(lib.yh_nchw_to_nhwc(hiplib.ptr(x), hiplib.ptr(xh), n, c, h, wi, c_p, c_p, F32, S), 'nchw_to_nhwc')
        packed = torch.empty(k * k * c_p, dtype=torch.float32, device=dev)
        bias = torch.empty(c_p, dtype=torch.float32, device=dev)
        cb = torch.zeros(c, dtype=torch.float32, device=dev) if b is None else b.detach().float().contiguous()
        hiplib.check(lib.yh_dw_pack_weights(F32, hiplib.ptr(w), hiplib.ptr(cb), None, None, None, None, 0.0, None, c, k, c_p,
                                            hiplib.ptr(packed), hiplib.ptr(bias), S), 'dw pack')
        y = torch.empty((n, ho, wo, c_p), dtype=torch.float32, device=dev)
        desc = hiplib.DwDesc(x=hiplib.ptr(xh), w=hiplib.ptr(packed), bias=hiplib.ptr(bias), y=hiplib.ptr(y), n=n, h=h, w_in=wi, c=c_p, ho=ho,
                             wo=wo, k=k, stride=s, pad=p, ldx=c_p, ldy=c_p, act=0, slope=0.0, dtype=F32)
        hiplib.check(lib.yh_dwconv2d_fwd(C.byref(desc), S), 'calibration depthwise conv')
    return y[..., :c].permute(0, 3, 1, 2)
