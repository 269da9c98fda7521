"""Device-agnostic drivers for single C-ABI entry points (used by CPU emulator tests and GPU tests).

Every helper takes torch tensors that live on the device the given ``lib`` executes on: the real
``libyolo_hip.so`` with CUDA tensors, or ``fakelib.FakeLib`` with CPU tensors.
"""
import ctypes as C

import torch

from engine import hiplib
from engine.hiplib import ConvDesc, StemDesc, PoolDesc, CopyDesc, AddDesc, DecodeDesc, DwDesc, SeDesc, QCopyDesc, QAddDesc

P = hiplib.ptr


def round_up(a, b):
    return (a + b - 1) // b * b


def tdtype(code):
    return torch.float16 if code == hiplib.YH_F16 else torch.float32


def stream():
    return hiplib.stream_ptr()


def pack_conv(lib, code, w, cb=None, bn=None, eps=1e-5, cmap=None, cin_phys=None):
    """w (cout,cin,k,k) fp32 on device; bn = (gamma, beta, mean, var) or None -> (packed, bias, cin_k, m_pad)."""
    cout, cin, kh, kw = w.shape
    kstep = 32 if code == hiplib.YH_F16 else 16
    cin_phys = cin_phys or round_up(cin, 8)
    cin_k = round_up(cin_phys, kstep)
    m_pad = round_up(round_up(cout, 8), 128)
    packed = torch.full((m_pad * kh * kw * cin_k,), 7.0, device=w.device, dtype=tdtype(code))
    bias = torch.full((m_pad,), 7.0, device=w.device, dtype=torch.float32)
    g, be, mu, var = bn if bn is not None else (None, None, None, None)
    rc = lib.yh_conv_pack_weights(code, P(w), P(cb), P(g), P(be), P(mu), P(var), eps, P(cmap), cout, cin, kh, kw, cin_k,
                                  m_pad, P(packed), P(bias), stream())
    assert rc == 0, rc
    return packed, bias, cin_k, m_pad


def conv(lib, code, x, packed, bias, cin_k, m_pad, cout_phys, k, stride, pad, act=0, slope=0.1, res=None, ups=1,
         out_f32=False, tile=0, cin=None, x_off=0, y=None, y_off=0, res_off=0, stats=None, bwd=None):
    """x: (N,H,W,ldx) NHWC buffer; reads channels [x_off, x_off+cin).  Returns the (N,Ho*ups,Wo*ups,ldy) output."""
    N, H, W, ldx = x.shape
    cin = cin if cin is not None else ldx - x_off
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    odt = torch.float32 if out_f32 else tdtype(code)
    if y is None:
        y = torch.full((N, Ho * ups, Wo * ups, cout_phys), 3.0, device=x.device, dtype=odt)
    d = ConvDesc(x=P(x, x_off), w=P(packed), bias=P(bias), res=None if res is None else P(res, res_off), y=P(y, y_off),
                 n=N, h=H, w_in=W, cin=cin, ho=Ho, wo=Wo, cout=cout_phys, kh=k, kw=k, stride=stride, pad=pad,
                 ldx=ldx, ldr=0 if res is None else res.shape[3], ldy=y.shape[3], cin_k=cin_k, m_pad=m_pad, act=act,
                 slope=slope, ups=ups, out_f32=1 if out_f32 else 0, dtype=code, tile=tile)
    if stats is not None:   # fused BatchNorm statistics: stats is a dict that receives the partial rows
        rows = int(lib.yh_conv2d_stats_rows(C.byref(d)))
        assert rows > 0
        ws = torch.full((rows * 2 * cout_phys,), float('nan'), device=x.device, dtype=torch.float32)
        d.stats_ws, d.stats_ws_floats = P(ws), ws.numel()
        stats['rows'], stats['ws'] = rows, ws
    if bwd is not None:     # the block whose gradient this launch completes (yh_conv_desc.bwd_z): dict(z, gamma, beta, mean, invstd, act[, slope])
        d.bwd_z, d.bwd_ldz, d.bwd_act, d.bwd_slope = P(bwd['z']), bwd['z'].shape[3], bwd['act'], bwd.get('slope', 0.1)
        d.bwd_gamma, d.bwd_beta, d.bwd_mean, d.bwd_invstd = P(bwd['gamma']), P(bwd['beta']), P(bwd['mean']), P(bwd['invstd'])
        rows = int(lib.yh_conv2d_bwd_stats_rows(C.byref(d)))
        bwd['rows'] = rows
        if rows <= 0:
            return None
        ws = torch.full((rows * 2 * cout_phys,), float('nan'), device=x.device, dtype=torch.float32)
        d.stats_ws, d.stats_ws_floats = P(ws), ws.numel()
        bwd['ws'] = ws
    rc = lib.yh_conv2d_fwd(C.byref(d), stream())
    assert rc == 0, 'yh_conv2d_fwd rc=%d' % rc
    return y


def stem(lib, code, x, w, cb=None, bn=None, eps=1e-5, stride=1, pad=1, act=1, slope=0.1, stats=None):
    cout, cin, kh, kw = w.shape
    N, _, H, W = x.shape
    cout_phys = round_up(cout, 8)
    cout_pad = cout if cout % 32 == 0 else round_up(cout, 16)
    packed = torch.empty(kh * kw * cin * cout_pad, device=x.device, dtype=torch.float32)
    bias = torch.empty(cout_pad, device=x.device, dtype=torch.float32)
    g, be, mu, var = bn if bn is not None else (None, None, None, None)
    rc = lib.yh_stem_pack_weights(P(w), P(cb), P(g), P(be), P(mu), P(var), eps, cout, cin, kh, kw, cout_pad, P(packed),
                                  P(bias), stream())
    assert rc == 0, rc
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    y = torch.full((N, Ho, Wo, cout_phys), 3.0, device=x.device, dtype=tdtype(code))
    d = StemDesc(x=P(x), w=P(packed), bias=P(bias), y=P(y), n=N, cin=cin, h=H, w_in=W, ho=Ho, wo=Wo, cout=cout_phys,
                 cout_pad=cout_pad, kh=kh, kw=kw, stride=stride, pad=pad, ldy=cout_phys, act=act, slope=slope, dtype=code)
    if stats is not None:      # the statistics epilogue of the MFMA first-layer kernel: rows of [2][cout] partial sums
        rows = int(lib.yh_conv2d_stem_stats_rows(C.byref(d)))
        assert rows > 0, 'this stem shape has no statistics epilogue'
        ws = torch.full((rows * 2 * cout_phys,), float('nan'), device=x.device, dtype=torch.float32)
        d.stats_ws, d.stats_ws_floats = P(ws), ws.numel()
        stats['rows'], stats['ws'] = rows, ws
    rc = lib.yh_conv2d_stem_fwd(C.byref(d), stream())
    assert rc == 0, rc
    return y


def maxpool(lib, code, x, k, stride, c=None, x_off=0):
    N, H, W, ldx = x.shape
    c = c if c is not None else ldx - x_off
    if k == 2 and stride == 1:
        Ho, Wo, pad_lo, edge_zero = H, W, 0, 1
    else:
        pad_lo, edge_zero = (k - 1) // 2, 0
        Ho, Wo = (H + 2 * pad_lo - k) // stride + 1, (W + 2 * pad_lo - k) // stride + 1
    y = torch.full((N, Ho, Wo, c), 3.0, device=x.device, dtype=x.dtype)
    d = PoolDesc(x=P(x, x_off), y=P(y), n=N, h=H, w_in=W, c=c, ho=Ho, wo=Wo, k=k, stride=stride, pad_lo=pad_lo,
                 edge_zero=edge_zero, ldx=ldx, ldy=c, dtype=code)
    rc = lib.yh_maxpool2d_fwd(C.byref(d), stream())
    assert rc == 0, rc
    return y


def copy_channels(lib, code, x, y, c, ups=1, x_off=0, y_off=0):
    N, H, W, ldx = x.shape
    d = CopyDesc(x=P(x, x_off), y=P(y, y_off), n=N, h=H, w_in=W, c=c, ups=ups, ldx=ldx, ldy=y.shape[3], dtype=code)
    rc = lib.yh_copy_channels(C.byref(d), stream())
    assert rc == 0, rc
    return y


def add_channels(lib, code, a, b, c, amap=None, bmap=None):
    """y = a + b over c channels; with amap / bmap (lists of c source channels, -1 = zero) the gather form."""
    N, H, W, lda = a.shape
    y = torch.full((N, H, W, c), 3.0, device=a.device, dtype=a.dtype)
    maps = [None if m is None else torch.tensor(m, dtype=torch.int32).to(a.device) for m in (amap, bmap)]
    d = AddDesc(a=P(a), b=P(b), y=P(y), pixels=N * H * W, c=c, lda=lda, ldb=b.shape[3], ldy=c, dtype=code,
                amap=P(maps[0]) if maps[0] is not None else None, bmap=P(maps[1]) if maps[1] is not None else None)
    rc = lib.yh_add_channels(C.byref(d), stream())
    assert rc == 0, rc
    if a.is_cuda:
        torch.cuda.synchronize()   # the maps must outlive the launch
    return y


def decode(lib, p, na, no, stride, anchor_vec, rows_total=None, row_off=0, want_raw=True):
    """p: (N,ny,nx,ldp) fp32 head map."""
    N, ny, nx, ldp = p.shape
    rows = na * ny * nx
    rows_total = rows_total or rows
    io = torch.full((N, rows_total, no), -1.0, device=p.device, dtype=torch.float32)
    raw = torch.full((N, na, ny, nx, no), -1.0, device=p.device, dtype=torch.float32) if want_raw else None
    d = DecodeDesc(p=P(p), io=P(io), raw=P(raw), n=N, ny=ny, nx=nx, na=na, no=no, ldp=ldp, rows_total=rows_total,
                   row_off=row_off, stride=float(stride))
    for a in range(na):
        d.anchor_w[a] = float(anchor_vec[a][0])
        d.anchor_h[a] = float(anchor_vec[a][1])
    rc = lib.yh_yolo_decode(C.byref(d), stream())
    assert rc == 0, rc
    return io, raw


def dwconv(lib, code, x, w, bn=None, cb=None, eps=1e-5, stride=1, act=3, slope=0.1, cmap=None, c_phys=None, x_off=0):
    """x: (N,H,W,ldx) NHWC; w: (C,1,k,k) fp32 on the same device; depthwise over channels [x_off, x_off + c_phys)."""
    N, H, W, ldx = x.shape
    c, _, k, _ = w.shape
    c_phys = c_phys or round_up(c, 8)
    pad = (k - 1) // 2
    packed = torch.full((k * k * c_phys,), 7.0, device=x.device, dtype=tdtype(code))
    bias = torch.full((c_phys,), 7.0, device=x.device, dtype=torch.float32)
    g, be, mu, var = bn if bn is not None else (None, None, None, None)
    rc = lib.yh_dw_pack_weights(code, P(w), P(cb), P(g), P(be), P(mu), P(var), eps, P(cmap), c, k, c_phys, P(packed), P(bias),
                                stream())
    assert rc == 0, rc
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    y = torch.full((N, Ho, Wo, c_phys), 3.0, device=x.device, dtype=x.dtype)
    d = DwDesc(x=P(x, x_off), w=P(packed), bias=P(bias), y=P(y), n=N, h=H, w_in=W, c=c_phys, ho=Ho, wo=Wo, k=k, stride=stride,
               pad=pad, ldx=ldx, ldy=c_phys, act=act, slope=slope, dtype=code)
    rc = lib.yh_dwconv2d_fwd(C.byref(d), stream())
    assert rc == 0, rc
    return y, packed, bias


def se(lib, code, x, w1, w2, c, cmap=None):
    """x: (N,H,W,c_phys) NHWC; w1 (cr,c), w2 (c,cr) fp32."""
    N, H, W, c_phys = x.shape
    pooled = torch.empty((N, c_phys), device=x.device, dtype=torch.float32)
    gate = torch.empty((N, c_phys), device=x.device, dtype=torch.float32)
    y = torch.full_like(x, 3.0)
    d = SeDesc(x=P(x), y=P(y), w1=P(w1), w2=P(w2), pooled=P(pooled), gate=P(gate), ch_map=P(cmap), n=N, h=H, w_in=W, c=c,
               c_phys=c_phys, cr=w1.shape[0], ldx=c_phys, ldy=c_phys, dtype=code)
    rc = lib.yh_se_fwd(C.byref(d), stream())
    assert rc == 0, rc
    return y, gate


# ------------------------------------------------------------------------------------------------ int8 (PTQ eval)
def qconv(lib, x, qw, w_scale, qbias, acc_scale, out_scale, k, stride, pad, act=1, slope=0.1, ups=1, out_f32=False, tile=0,
          cin=None, x_off=0, y=None, y_off=0, res=None, qadd=None):
    """x int8 (N,H,W,ldx); qw fp32 (cout,cin,k,k) = int grid * w_scale; qbias fp32 (cout,) real units."""
    N, H, W, ldx = x.shape
    cout, cin_l = qw.shape[0], qw.shape[1]
    cin = cin if cin is not None else ldx - x_off
    cin_k = round_up(cin, 64)
    cout_phys = round_up(cout, 16)
    m_pad = round_up(cout_phys, 128)
    packed = torch.full((m_pad * k * k * cin_k,), 7, device=x.device, dtype=torch.int8)
    rc = lib.yh_qconv_pack_weights(P(qw), float(w_scale), None, cout, cin_l, k, k, cin_k, m_pad, P(packed), stream())
    assert rc == 0, rc
    bias = torch.zeros(m_pad, device=x.device, dtype=torch.float32)
    bias[:cout] = qbias
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    if y is None:
        y = torch.full((N, Ho * ups, Wo * ups, cout_phys), 3, device=x.device, dtype=torch.float32 if out_f32 else torch.int8)
    d = ConvDesc(x=P(x, x_off), w=P(packed), bias=P(bias), res=None, y=P(y, y_off), n=N, h=H, w_in=W, cin=cin, ho=Ho, wo=Wo,
                 cout=cout_phys, kh=k, kw=k, stride=stride, pad=pad, ldx=ldx, ldr=0, ldy=y.shape[3], cin_k=cin_k, m_pad=m_pad,
                 act=act, slope=slope, ups=ups, out_f32=1 if out_f32 else 0, dtype=hiplib.YH_I8, tile=tile,
                 acc_scale=float(acc_scale), out_scale=float(out_scale))
    if res is not None:    # fused quantised shortcut: res int8 (N, Ho, Wo, ldr), qadd = (q_rx, q_ra, q_scale_x, q_scale_a, q_inv_scale_sum)
        d.res, d.ldr = P(res), res.shape[3]
        d.q_rx, d.q_ra, d.q_scale_x, d.q_scale_a, d.q_inv_scale_sum = (float(v) for v in qadd)
    rc = lib.yh_conv2d_fwd(C.byref(d), stream())
    assert rc == 0, 'yh_conv2d_fwd(i8) rc=%d' % rc
    qconv.last_tile = int(lib.yh_conv2d_tile(C.byref(d))) if hasattr(lib, 'yh_conv2d_tile') else -1      # which kernel the call took
    return y, packed


def qcopy(lib, x, y, c, ratio=1.0, ups=1, x_off=0, y_off=0):
    N, H, W, ldx = x.shape
    d = QCopyDesc(x=P(x, x_off), y=P(y, y_off), n=N, h=H, w_in=W, c=c, ups=ups, ldx=ldx, ldy=y.shape[3], ratio=float(ratio))
    rc = lib.yh_qcopy(C.byref(d), stream())
    assert rc == 0, rc
    return y


def qpool(lib, x, k, stride):
    N, H, W, c = x.shape
    if k == 2 and stride == 1:
        Ho, Wo, pad_lo, edge_zero = H, W, 0, 1
    else:
        pad_lo, edge_zero = (k - 1) // 2, 0
        Ho, Wo = (H + 2 * pad_lo - k) // stride + 1, (W + 2 * pad_lo - k) // stride + 1
    y = torch.full((N, Ho, Wo, c), 3, device=x.device, dtype=torch.int8)
    d = PoolDesc(x=P(x), y=P(y), n=N, h=H, w_in=W, c=c, ho=Ho, wo=Wo, k=k, stride=stride, pad_lo=pad_lo, edge_zero=edge_zero,
                 ldx=c, ldy=c, dtype=hiplib.YH_I8)
    rc = lib.yh_qpool(C.byref(d), stream())
    assert rc == 0, rc
    return y


def qadd(lib, x, a, rx, ra, scale_x, scale_a, scale_sum):
    N, H, W, c = x.shape
    y = torch.full((N, H, W, c), 3, device=x.device, dtype=torch.int8)
    d = QAddDesc(x=P(x), a=P(a), y=P(y), pixels=N * H * W, c=c, ldx=c, lda=a.shape[3], ldy=c, rx=float(rx), ra=float(ra),
                 scale_x=float(scale_x), scale_a=float(scale_a), inv_scale_sum=float(1.0 / scale_sum))
    rc = lib.yh_qadd(C.byref(d), stream())
    assert rc == 0, rc
    return y


# ------------------------------------------------------------------------------------------ training path
from engine.hiplib import BnDesc, WgradDesc, ResampleDesc, CastDesc  # noqa: E402


def with_bn_workspace(lib, d):
    """Attach the two-stage reduction workspace the library asks for (kept alive on the descriptor object)."""
    need = int(lib.yh_bn_reduce_workspace(C.byref(d)))
    if need:
        d._ws_keep = torch.full((need,), float('nan'), device='cuda', dtype=torch.float32)
        d.ws, d.ws_floats = P(d._ws_keep), need
    return d


def bn_desc(code, z, c, ldz_off=0, dy=None, res=None, out=None, gamma=None, beta=None, mean=None, invstd=None, s1=None,
            s2=None, rmean=None, rvar=None, act=1, slope=0.1, ups=1, eps=1e-5, momentum=0.1, out_off=0):
    """z, dy, res, out: (N,H,W,ld) NHWC buffers (channels [off, off+c) are used)."""
    N, H, W, ldz = z.shape
    return BnDesc(z=P(z, ldz_off), dy=P(dy), res=P(res), out=P(out, out_off) if out is not None else None, gamma=P(gamma),
                  beta=P(beta), mean=P(mean), invstd=P(invstd), sum=P(s1), sumsq=P(s2), running_mean=P(rmean),
                  running_var=P(rvar), pixels=N * H * W, n=N, h=H, w_in=W, c=c, ldz=ldz,
                  lddy=0 if dy is None else dy.shape[3], ldr=0 if res is None else res.shape[3],
                  ldo=0 if out is None else out.shape[3], act=act, ups=ups, dtype=code, slope=slope, eps=eps,
                  momentum=momentum)


def call(lib, name, desc):
    rc = getattr(lib, name)(C.byref(desc), stream())
    assert rc == 0, '%s rc=%d' % (name, rc)


def wgrad(lib, code, x, dz, cin, cout, k, stride, pad, x_off=0, dz_off=0, splits=0, use_ws=True):
    """x (N,H,W,ldx), dz (N,Ho,Wo,lddz) -> dw (cout,cin,k,k) fp32 accumulated from zero."""
    N, H, W, ldx = x.shape
    _, Ho, Wo, lddz = dz.shape
    dw = torch.zeros((cout, cin, k, k), device=x.device, dtype=torch.float32)
    d = WgradDesc(x=P(x, x_off), dz=P(dz, dz_off), dw=P(dw), n=N, h=H, w_in=W, cin=cin, ho=Ho, wo=Wo, cout=cout, kh=k, kw=k,
                  stride=stride, pad=pad, ldx=ldx, lddz=lddz, dtype=code, splits=splits)
    need = int(lib.yh_conv2d_wgrad_workspace(C.byref(d))) if use_ws else 0
    if need:   # two-stage reduction (per-split partial tiles + summing launch); without it: fp32 atomics
        ws = torch.full((need,), float('nan'), device=x.device, dtype=torch.float32)
        d.ws, d.ws_floats = P(ws), need
    call(lib, 'yh_conv2d_wgrad', d)
    return dw


def stem_wgrad(lib, code, x, dz, cout, stride=1, pad=1):
    N, cin, H, W = x.shape
    _, Ho, Wo, lddz = dz.shape
    dw = torch.zeros((cout, cin, 3, 3), device=x.device, dtype=torch.float32)
    d = WgradDesc(x=P(x), dz=P(dz), dw=P(dw), n=N, h=H, w_in=W, cin=cin, ho=Ho, wo=Wo, cout=cout, kh=3, kw=3, stride=stride,
                  pad=pad, ldx=0, lddz=lddz, dtype=code, splits=0)
    call(lib, 'yh_stem_wgrad', d)
    return dw


def dgrad(lib, code, dz, w, in_hw, stride, pad, acc=None, fused=False):
    """Data gradient through yh_conv2d_fwd on the dgrad weight image (stride 2: four parity phases, ups = 3 scatter, or with
    ``fused`` all four in one 2x2-tap pass, ups = 4).
    dz (N,Ho,Wo,cout_phys), w (cout,cin,k,k) fp32 -> dx (N,H,W,cin_phys); ``acc`` (same shape) is accumulated into."""
    cout, cin, k, _ = w.shape
    N, Ho, Wo, cphys = dz.shape
    H, W = in_hw
    kstep = 32 if code == hiplib.YH_F16 else 16
    cin_phys = round_up(cin, 8)
    cout_k, dm_pad = round_up(cphys, kstep), round_up(cin_phys, 128)
    wt = torch.empty(dm_pad * k * k * cout_k, device=dz.device, dtype=tdtype(code))
    rc = lib.yh_conv_pack_weights_dgrad(code, P(w), cout, cin, k, k, cout_k, dm_pad, P(wt), stream())
    assert rc == 0, rc
    dx = acc if acc is not None else torch.full((N, H, W, cin_phys), 3.0, device=dz.device, dtype=dz.dtype)
    zero_bias = torch.zeros(dm_pad, device=dz.device, dtype=torch.float32)
    common = dict(x=P(dz), bias=P(zero_bias), res=P(acc), y=P(dx), n=N, h=Ho, w_in=Wo, cin=cphys, cout=cin_phys, stride=1,
                  ldx=cphys, ldr=0 if acc is None else cin_phys, ldy=cin_phys, cin_k=cout_k, m_pad=dm_pad, act=0, slope=0.0,
                  out_f32=0, dtype=code, tile=0)
    if stride == 2 and fused:
        from engine.hiplib import PackItem
        fm_pad = round_up(4 * cin_phys, 128)
        img = torch.empty(fm_pad * 4 * cout_k, device=dz.device, dtype=tdtype(code))
        item = PackItem(w=P(w), packed=P(img), mode=5, dtype=code, cout=cout, cin=cin, kh=k, kw=k, k_pad=cout_k, m_pad=fm_pad,
                        pad=pad, cout_pad=cin_phys)
        table = torch.frombuffer(bytearray(bytes(item)), dtype=torch.uint8).to(dz.device)
        rc = lib.yh_pack_batch(P(table), 1, stream())
        assert rc == 0, rc
        zb = torch.zeros(fm_pad, device=dz.device, dtype=torch.float32)
        call(lib, 'yh_conv2d_fwd', ConvDesc(w=P(img), ho=(H + 1) // 2, wo=(W + 1) // 2, kh=2, kw=2, pad=0, ups=4, y_h=H, y_w=W,
                                            **dict(common, bias=P(zb), cout=4 * cin_phys, m_pad=fm_pad)))
        if dz.is_cuda:
            torch.cuda.synchronize()   # the pack table and image must outlive the launches
        return dx
    if stride == 2:   # four parity phases, scattered onto every other pixel
        keep = []
        for a in (0, 1):
            for b in (0, 1):
                khp, kwp = C.c_int(0), C.c_int(0)
                img = torch.empty(dm_pad * 4 * cout_k, device=dz.device, dtype=tdtype(code))
                rc = lib.yh_conv_pack_weights_dgrad_phase(code, P(w), cout, cin, k, k, pad, a, b, cout_k, dm_pad, P(img),
                                                          C.byref(khp), C.byref(kwp), stream())
                assert rc == 0, rc
                keep.append(img)
                hp, wp = (H - a + 1) // 2, (W - b + 1) // 2
                if hp <= 0 or wp <= 0:
                    continue
                call(lib, 'yh_conv2d_fwd', ConvDesc(w=P(img), ho=hp, wo=wp, kh=khp.value, kw=kwp.value, pad=0, ups=3, y_h=H, y_w=W,
                                                    y_off_h=a, y_off_w=b, **common))
        return dx
    d = ConvDesc(w=P(wt), ho=H, wo=W, kh=k, kw=k, pad=k - 1 - pad, ups=1, **common)
    call(lib, 'yh_conv2d_fwd', d)
    return dx


def stem_wgrad_mfma(lib, code, x, dz, cout, stride=1, pad=1):
    """First-layer weight gradient the way the engine runs it: image -> NHWC dtype with 8 channels, MFMA wgrad, cin_w."""
    N, cin, H, W = x.shape
    _, Ho, Wo, lddz = dz.shape
    img = torch.full((N, H, W, 8), 5.0, device=x.device, dtype=tdtype(code))
    rc = lib.yh_nchw_to_nhwc(P(x), P(img), N, cin, H, W, 8, 8, code, stream())
    assert rc == 0, rc
    dw = torch.zeros((cout, cin, 3, 3), device=x.device, dtype=torch.float32)
    d = WgradDesc(x=P(img), dz=P(dz), dw=P(dw), n=N, h=H, w_in=W, cin=8, ho=Ho, wo=Wo, cout=cout, kh=3, kw=3, stride=stride,
                  pad=pad, ldx=8, lddz=lddz, dtype=code, splits=0, cin_w=cin)
    need = int(lib.yh_conv2d_wgrad_workspace(C.byref(d)))
    if need:
        ws = torch.full((need,), float('nan'), device=x.device, dtype=torch.float32)
        d.ws, d.ws_floats = P(ws), need
    call(lib, 'yh_conv2d_wgrad', d)
    return dw, img


def stem_bwd(lib, x, dy, z, gamma, beta, mean, invstd, act=1, slope=0.1, dz1=None, w1=None):
    """yh_stem_bwd on x (N,cin,H,W) fp32, dy / z (N,H,W,ld) f16: returns (dw [c][cin][3][3], dgamma, dbeta), all fp32.
    With dz1 (N,H1,W1,ld1) f16 and w1 (64,32,3,3) fp32 the next conv's data gradient is fused in and dy is ignored."""
    from engine.hiplib import StemBwdDesc
    N, cin, H, W = x.shape
    c = gamma.numel()
    dev = x.device
    dw = torch.zeros(c, cin, 3, 3, device=dev)
    dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
    d = StemBwdDesc(x=P(x), dy=P(dy), z=P(z), gamma=P(gamma), beta=P(beta), mean=P(mean), invstd=P(invstd), dgamma=P(dg), dbeta=P(db),
                    dw=P(dw), n=N, cin=cin, h=H, w_in=W, cout=c, lddy=0 if dy is None else dy.shape[3], ldz=z.shape[3], act=act, slope=slope)
    keep = None
    if dz1 is not None:
        k1 = w1.shape[0]
        img = torch.zeros(128 * 9 * k1, device=dev, dtype=torch.float16)
        rc = lib.yh_conv_pack_weights_dgrad(hiplib.YH_F16, P(w1), k1, c, 3, 3, k1, 128, P(img), stream())
        assert rc == 0, rc
        d.dz1, d.w1, d.h1, d.w1_in, d.k1, d.k1_pad, d.lddz1 = P(dz1), P(img), dz1.shape[1], dz1.shape[2], k1, k1, dz1.shape[3]
        keep = img
    need = int(lib.yh_stem_bwd_workspace(C.byref(d)))
    assert need > 0
    ws = torch.full((need,), float('nan'), device=dev)
    d.ws, d.ws_floats = P(ws), need
    rc = lib.yh_stem_bwd(C.byref(d), stream())
    assert rc == 0, rc
    if x.is_cuda:
        torch.cuda.synchronize()
    return dw, dg, db
