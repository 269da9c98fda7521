"""Host emulation of the libyolo_hip.so C ABI — TEST INFRASTRUCTURE ONLY.

``FakeLib`` exposes the same entry points as the ctypes handle returned by ``engine.hiplib.load()`` and
executes every descriptor on *host* memory with torch CPU ops, reading and writing the raw addresses in
the descriptors exactly like the kernels do (pitches, channel offsets, padded weight images, slots and
fix-ups).  Injecting it into ``DarknetEngine`` lets the CPU-only test tier validate the graph lowering
(fusion decisions, zero-copy concat placement, channel maps, weight packing layout, plan replay) against
the reference goldens without a GPU.  The kernels themselves are validated on the GPU tier against the
same oracle.  Product code never imports this module.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.nn.functional as F

from engine import hiplib
from engine.hiplib import (StemBwdDesc, ConvDesc, StemDesc, PoolDesc, CopyDesc, AddDesc, DecodeDesc, DwDesc, SeDesc, QCopyDesc, QPoolDesc,
                           QAddDesc, BnStatsDesc, BnFinalizeDesc, BnActFwdDesc, BnBwdReduceDesc, BnBwdApplyDesc, WgradDesc,
                           StemWgradDesc, DilateDesc, UpsampleBwdDesc, CastDesc, LayoutDesc, PoolBwdDesc, PackBatchDesc, PackItem, DwWgradDesc, DwDgradDesc, SeBwdDesc)

_NP = {hiplib.YH_F16: np.float16, hiplib.YH_F32: np.float32, hiplib.YH_I8: np.int8}
_DESC = {hiplib.OP_CONV: ConvDesc, hiplib.OP_STEM: StemDesc, hiplib.OP_POOL: PoolDesc, hiplib.OP_COPY: CopyDesc,
         hiplib.OP_ADD: AddDesc, hiplib.OP_DECODE: DecodeDesc, hiplib.OP_DW: DwDesc, hiplib.OP_SE: SeDesc,
         hiplib.OP_QCOPY: QCopyDesc, hiplib.OP_QPOOL: QPoolDesc, hiplib.OP_QADD: QAddDesc,
         hiplib.OP_BN_STATS: BnStatsDesc, hiplib.OP_BN_FINALIZE: BnFinalizeDesc, hiplib.OP_BN_ACT_FWD: BnActFwdDesc,
         hiplib.OP_BN_BWD_REDUCE: BnBwdReduceDesc, hiplib.OP_BN_BWD_APPLY: BnBwdApplyDesc, hiplib.OP_WGRAD: WgradDesc,
         hiplib.OP_STEM_WGRAD: StemWgradDesc, hiplib.OP_DILATE2: DilateDesc, hiplib.OP_UPSAMPLE2_BWD: UpsampleBwdDesc,
         hiplib.OP_CAST_F32: CastDesc, hiplib.OP_NCHW_TO_NHWC: LayoutDesc, hiplib.OP_POOL_BWD: PoolBwdDesc, hiplib.OP_PACK_BATCH: PackBatchDesc, hiplib.OP_DW_WGRAD: DwWgradDesc, hiplib.OP_DW_DGRAD: DwDgradDesc,
         hiplib.OP_SE_BWD: SeBwdDesc, hiplib.OP_STEM_BWD: StemBwdDesc}


def _addr(p):
    if p is None:
        return 0
    if isinstance(p, C.c_void_p):
        return p.value or 0
    return int(p)


def flat(p, n, dtype):
    """numpy view of n elements of dtype at host address p."""
    a = _addr(p)
    assert a != 0, 'null pointer dereferenced by the emulator'
    itemsize = np.dtype(dtype).itemsize
    buf = (C.c_char * (n * itemsize)).from_address(a)
    return np.frombuffer(buf, dtype=dtype, count=n)


def pitched(p, pixels, c, ld, dtype):
    """[pixels, c] view with row pitch ld (elements) at host address p."""
    base = flat(p, (pixels - 1) * ld + c, dtype)
    es = base.itemsize
    return np.lib.stride_tricks.as_strided(base, shape=(pixels, c), strides=(ld * es, es), writeable=True)


def rnd_away(t):
    """PTQ rounding: half away from zero (quantized_ptq_cos.py:14-20)."""
    return torch.sign(t) * torch.floor(torch.abs(t) + 0.5)


def _act(v, act, slope):
    if act == 1:
        return torch.where(v > 0, v, v * slope)
    if act == 2:
        return v.clamp(min=0)
    if act == 3:
        return v.clamp(0, 6)
    if act == 4:
        return v * (v + 3).clamp(0, 6) / 6
    if act == 5:
        return v * torch.tanh(F.softplus(v))
    return v


class FakeLib:
    def __init__(self):
        self.plans = {}
        self.next = 1
        self.calls = []

    # ---- misc
    def yh_abi_version(self):
        return 1

    def yh_error_string(self, rc):
        return ('fake error %d' % rc).encode()

    # ---- packing
    @staticmethod
    def _fold(w, cb, g, be, mu, var, eps, cout):
        scale = torch.ones(cout)
        bias = torch.zeros(cout) if cb is None else cb.clone()
        if g is not None:
            sd = torch.sqrt(var + eps)
            scale = g / sd
            bias = (be - g * mu / sd) + bias * scale
        return w * scale.view(-1, 1, 1, 1), bias

    def yh_conv_pack_weights(self, dtype, w, cb, g, be, mu, var, eps, cmap, cout, cin, kh, kw, cin_k, m_pad, packed,
                             bias_out, stream):
        t = lambda p, n: None if not _addr(p) else torch.from_numpy(flat(p, n, np.float32).copy())
        W = t(w, cout * cin * kh * kw).view(cout, cin, kh, kw)
        Wf, b = self._fold(W, t(cb, cout), t(g, cout), t(be, cout), t(mu, cout), t(var, cout), eps, cout)
        m = np.arange(cin) if not _addr(cmap) else flat(cmap, cin, np.int32).copy()
        assert cin_k % (32 if dtype == hiplib.YH_F16 else 16) == 0 and m.max() < cin_k and m_pad % 128 == 0
        img = torch.zeros(m_pad, kh * kw, cin_k)
        img[:cout][:, :, torch.from_numpy(m).long()] = Wf.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
        flat(packed, m_pad * kh * kw * cin_k, _NP[dtype])[:] = img.reshape(-1).numpy().astype(_NP[dtype])
        bo = flat(bias_out, m_pad, np.float32)
        bo[:] = 0
        bo[:cout] = b.numpy()
        return 0

    def yh_stem_pack_weights(self, w, cb, g, be, mu, var, eps, cout, cin, kh, kw, cout_pad, packed, bias_out, stream):
        t = lambda p, n: None if not _addr(p) else torch.from_numpy(flat(p, n, np.float32).copy())
        W = t(w, cout * cin * kh * kw).view(cout, cin, kh, kw)
        Wf, b = self._fold(W, t(cb, cout), t(g, cout), t(be, cout), t(mu, cout), t(var, cout), eps, cout)
        img = torch.zeros(kh * kw, cin, cout_pad)
        img[:, :, :cout] = Wf.permute(2, 3, 1, 0).reshape(kh * kw, cin, cout)
        flat(packed, kh * kw * cin * cout_pad, np.float32)[:] = img.reshape(-1).numpy()
        bo = flat(bias_out, cout_pad, np.float32)
        bo[:] = 0
        bo[:cout] = b.numpy()
        return 0

    def yh_dw_pack_weights(self, dtype, w, cb, g, be, mu, var, eps, cmap, c, k, c_phys, packed, bias_out, stream):
        t = lambda p, n: None if not _addr(p) else torch.from_numpy(flat(p, n, np.float32).copy())
        W = t(w, c * k * k).view(c, 1, k, k)
        Wf, b = self._fold(W, t(cb, c), t(g, c), t(be, c), t(mu, c), t(var, c), eps, c)
        m = torch.arange(c) if not _addr(cmap) else torch.from_numpy(flat(cmap, c, np.int32).copy()).long()
        img = torch.zeros(k * k, c_phys)
        img[:, m] = Wf.view(c, k * k).t()
        flat(packed, k * k * c_phys, _NP[dtype])[:] = img.reshape(-1).numpy().astype(_NP[dtype])
        bo = flat(bias_out, c_phys, np.float32)
        bo[:] = 0
        bo[m.numpy()] = b.numpy()
        return 0

    # ---- ops
    def yh_dwconv2d_fwd(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        x = torch.from_numpy(pitched(d.x, d.n * d.h * d.w_in, d.c, d.ldx, npdt).astype(np.float32))
        x = x.view(d.n, d.h, d.w_in, d.c).permute(0, 3, 1, 2)
        w = torch.from_numpy(flat(d.w, d.k * d.k * d.c, npdt).astype(np.float32)).view(d.k, d.k, d.c).permute(2, 0, 1)
        b = torch.from_numpy(flat(d.bias, d.c, np.float32).copy())
        y = _act(F.conv2d(x, w.unsqueeze(1).contiguous(), b, stride=d.stride, padding=d.pad, groups=d.c), d.act, d.slope)
        assert y.shape[2] == d.ho and y.shape[3] == d.wo
        pitched(d.y, d.n * d.ho * d.wo, d.c, d.ldy, npdt)[:] = y.permute(0, 2, 3, 1).reshape(-1, d.c).numpy().astype(npdt)
        return 0

    def yh_se_fwd(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        hw = d.h * d.w_in
        x = torch.from_numpy(pitched(d.x, d.n * hw, d.c_phys, d.ldx, npdt).astype(np.float32)).view(d.n, hw, d.c_phys)
        m = torch.arange(d.c) if not _addr(d.ch_map) else torch.from_numpy(flat(d.ch_map, d.c, np.int32).copy()).long()
        w1 = torch.from_numpy(flat(d.w1, d.cr * d.c, np.float32).copy()).view(d.cr, d.c)
        w2 = torch.from_numpy(flat(d.w2, d.c * d.cr, np.float32).copy()).view(d.c, d.cr)
        pooled = x.mean(1)
        hidden = (pooled[:, m] @ w1.t()).clamp(min=0)
        gate = torch.zeros(d.n, d.c_phys)
        gate[:, m] = ((hidden @ w2.t()) + 3).clamp(0, 6) / 6
        flat(d.pooled, d.n * d.c_phys, np.float32)[:] = pooled.reshape(-1).numpy()
        flat(d.gate, d.n * d.c_phys, np.float32)[:] = gate.reshape(-1).numpy()
        y = x * gate[:, None, :]
        pitched(d.y, d.n * hw, d.c_phys, d.ldy, npdt)[:] = y.reshape(-1, d.c_phys).numpy().astype(npdt)
        return 0

    def yh_qconv_pack_weights(self, qw, w_scale, cmap, cout, cin, kh, kw, cin_k, m_pad, packed, stream):
        W = torch.from_numpy(flat(qw, cout * cin * kh * kw, np.float32).copy()).view(cout, cin, kh, kw)
        q = rnd_away(W / w_scale).clamp(-128, 127)
        m = np.arange(cin) if not _addr(cmap) else flat(cmap, cin, np.int32).copy()
        assert cin_k % 64 == 0 and m_pad % 128 == 0
        img = torch.zeros(m_pad, kh * kw, cin_k)
        img[:cout][:, :, torch.from_numpy(m).long()] = q.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
        flat(packed, m_pad * kh * kw * cin_k, np.int8)[:] = img.reshape(-1).numpy().astype(np.int8)
        return 0

    def _qconv(self, d):
        assert d.cin % 16 == 0 and d.ldx % 16 == 0 and d.cin_k % 64 == 0
        x = torch.from_numpy(pitched(d.x, d.n * d.h * d.w_in, d.cin, d.ldx, np.int8).astype(np.float64))
        x = x.view(d.n, d.h, d.w_in, d.cin).permute(0, 3, 1, 2)
        taps = d.kh * d.kw
        wimg = torch.from_numpy(flat(d.w, d.m_pad * taps * d.cin_k, np.int8).astype(np.float64)).view(d.m_pad, d.kh, d.kw, d.cin_k)
        w = wimg[:d.cout, :, :, :d.cin].permute(0, 3, 1, 2).contiguous()
        b = torch.from_numpy(flat(d.bias, d.m_pad, np.float32)[:d.cout].copy())
        acc = F.conv2d(x, w, None, stride=d.stride, padding=d.pad)  # exact integer sums in fp64
        y = (acc.float() * d.acc_scale + b.view(1, -1, 1, 1))
        y = _act(y, d.act, d.slope)
        q = rnd_away(y / d.out_scale).clamp(-128, 127)
        if _addr(d.res):   # fused quantised shortcut (yh_qadd on the value the conv would have stored)
            assert d.ups == 1 and not d.out_f32 and d.act in (0, 1, 5) and d.q_rx > 0 and d.q_inv_scale_sum > 0
            r = torch.from_numpy(pitched(d.res, d.n * d.ho * d.wo, d.cout, d.ldr, np.int8).astype(np.float32))
            r = r.view(d.n, d.ho, d.wo, d.cout).permute(0, 3, 1, 2)
            ssum = rnd_away(q * d.q_rx) * d.q_scale_x + rnd_away(r * d.q_ra) * d.q_scale_a
            q = rnd_away(ssum * d.q_inv_scale_sum).clamp(-128, 127)
        if d.ups == 2:
            q = q.repeat_interleave(2, 2).repeat_interleave(2, 3)
        if d.out_f32:
            out = pitched(d.y, d.n * q.shape[2] * q.shape[3], d.cout, d.ldy, np.float32)
            out[:] = (q * d.out_scale).permute(0, 2, 3, 1).reshape(-1, d.cout).numpy()
        else:
            out = pitched(d.y, d.n * q.shape[2] * q.shape[3], d.cout, d.ldy, np.int8)
            out[:] = q.permute(0, 2, 3, 1).reshape(-1, d.cout).numpy().astype(np.int8)
        return 0

    def yh_qcopy(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        x = torch.from_numpy(pitched(d.x, d.n * d.h * d.w_in, d.c, d.ldx, np.int8).astype(np.float32))
        if d.ratio != 1.0:
            x = rnd_away(x * d.ratio).clamp(-128, 127)
        x = x.view(d.n, d.h, d.w_in, d.c)
        if d.ups == 2:
            x = x.repeat_interleave(2, 1).repeat_interleave(2, 2)
        pitched(d.y, d.n * d.h * d.w_in * d.ups * d.ups, d.c, d.ldy, np.int8)[:] = x.reshape(-1, d.c).numpy().astype(np.int8)
        return 0

    def yh_qpool(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        x = torch.from_numpy(pitched(d.x, d.n * d.h * d.w_in, d.c, d.ldx, np.int8).astype(np.float32))
        x = x.view(d.n, d.h, d.w_in, d.c).permute(0, 3, 1, 2)
        if d.edge_zero:
            need_h = (d.ho - 1) * d.stride + d.k - d.h
            need_w = (d.wo - 1) * d.stride + d.k - d.w_in
            y = F.max_pool2d(F.pad(x, (0, max(need_w, 0), 0, max(need_h, 0)), value=0.0), d.k, d.stride, 0)
        else:
            y = F.max_pool2d(x, d.k, d.stride, d.pad_lo)
        pitched(d.y, d.n * d.ho * d.wo, d.c, d.ldy, np.int8)[:] = y.permute(0, 2, 3, 1).reshape(-1, d.c).numpy().astype(np.int8)
        return 0

    def yh_qadd(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        x = torch.from_numpy(pitched(d.x, d.pixels, d.c, d.ldx, np.int8).astype(np.float32))
        a = torch.from_numpy(pitched(d.a, d.pixels, d.c, d.lda, np.int8).astype(np.float32))
        s = rnd_away(x * d.rx) * d.scale_x + rnd_away(a * d.ra) * d.scale_a
        pitched(d.y, d.pixels, d.c, d.ldy, np.int8)[:] = rnd_away(s * d.inv_scale_sum).clamp(-128, 127).numpy().astype(np.int8)
        return 0

    def yh_conv2d_fwd(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        if d.dtype == hiplib.YH_I8:
            return self._qconv(d)
        npdt = _NP[d.dtype]
        vec, bk = (8, 32) if d.dtype == hiplib.YH_F16 else (4, 16)
        assert d.cin % vec == 0 and d.ldx % vec == 0 and d.cin_k % bk == 0 and d.cout % 4 == 0 and d.ldy % 4 == 0
        assert _addr(d.x) % 16 == 0 and _addr(d.w) % 16 == 0 and _addr(d.y) % 8 == 0
        x = torch.from_numpy(pitched(d.x, d.n * d.h * d.w_in, d.cin, d.ldx, npdt).astype(np.float32))
        x = x.view(d.n, d.h, d.w_in, d.cin).permute(0, 3, 1, 2)
        taps = d.kh * d.kw
        wimg = torch.from_numpy(flat(d.w, d.m_pad * taps * d.cin_k, npdt).astype(np.float32)).view(d.m_pad, d.kh, d.kw, d.cin_k)
        w = wimg[:d.cout, :, :, :d.cin].permute(0, 3, 1, 2).contiguous()
        assert wimg[:, :, :, d.cin:].abs().max().item() == 0 if d.cin_k > d.cin else True
        b = torch.from_numpy(flat(d.bias, d.m_pad, np.float32)[:d.cout].copy())
        if d.ups == 4:   # the four phases of a stride-2 data gradient at once: row group p = 2a + b -> pixel (2ho + a, 2wo + b)
            assert d.kh == 2 and d.kw == 2 and d.pad == 0 and d.stride == 1 and d.cout % 16 == 0
            cpp = d.cout // 4
            xp = F.pad(x, (0, max(d.wo + 1 - d.w_in, 0), 0, max(d.ho + 1 - d.h, 0)))
            y = _act(F.conv2d(xp, w, b)[:, :, :d.ho, :d.wo], d.act, d.slope)
            out = pitched(d.y, d.n * d.y_h * d.y_w, cpp, d.ldy, np.float32 if d.out_f32 else npdt)
            rr = pitched(d.res, d.n * d.y_h * d.y_w, cpp, d.ldr, npdt) if _addr(d.res) else None
            full = out.reshape(d.n, d.y_h, d.y_w, cpp)
            for ph in range(4):
                a_, b_ = ph >> 1, ph & 1
                part = y[:, ph * cpp:(ph + 1) * cpp].permute(0, 2, 3, 1)[:, :(d.y_h - a_ + 1) // 2, :(d.y_w - b_ + 1) // 2]
                if rr is not None:
                    part = part + torch.from_numpy(rr.reshape(d.n, d.y_h, d.y_w, cpp)[:, a_::2, b_::2].astype(np.float32))
                full[:, a_::2, b_::2] = part.numpy().astype(out.dtype)
            return 0
        if d.ups == 3:   # phase scatter: free window geometry, taps beyond the input read zeros
            need_h, need_w = (d.ho - 1) * d.stride + d.kh - d.pad, (d.wo - 1) * d.stride + d.kw - d.pad
            x = F.pad(x, (d.pad, max(need_w - d.w_in, 0), d.pad, max(need_h - d.h, 0)))
            y = F.conv2d(x, w, b, stride=d.stride)[:, :, :d.ho, :d.wo]
            y = _act(y, d.act, d.slope)
            npdt_o = np.float32 if d.out_f32 else npdt
            out = pitched(d.y, d.n * d.y_h * d.y_w, d.cout, d.ldy, npdt_o)
            idx = ((np.arange(d.n)[:, None, None] * d.y_h + 2 * np.arange(d.ho)[None, :, None] + d.y_off_h) * d.y_w
                   + 2 * np.arange(d.wo)[None, None, :] + d.y_off_w).reshape(-1)
            vals = y.permute(0, 2, 3, 1).reshape(-1, d.cout)
            if _addr(d.res):
                rr = pitched(d.res, d.n * d.y_h * d.y_w, d.cout, d.ldr, npdt)
                vals = vals + torch.from_numpy(rr[idx].astype(np.float32))
            out[idx] = vals.numpy().astype(npdt_o)
            return 0
        y = F.conv2d(x, w, b, stride=d.stride, padding=d.pad)
        assert y.shape[2] == d.ho and y.shape[3] == d.wo
        y = _act(y, d.act, d.slope)
        if _addr(d.res):
            r = torch.from_numpy(pitched(d.res, d.n * d.ho * d.wo, d.cout, d.ldr, npdt).astype(np.float32))
            y = y + r.view(d.n, d.ho, d.wo, d.cout).permute(0, 3, 1, 2)
        if d.ups == 2:
            y = y.repeat_interleave(2, 2).repeat_interleave(2, 3)
        odt = np.float32 if d.out_f32 else npdt
        out = pitched(d.y, d.n * y.shape[2] * y.shape[3], d.cout, d.ldy, odt)
        out[:] = y.permute(0, 2, 3, 1).reshape(-1, d.cout).numpy().astype(odt)
        if _addr(d.bwd_z):      # backward sums of the block this data gradient completes (yh_conv_desc.bwd_z): one partial row
            assert self.yh_conv2d_bwd_stats_rows(d) == 1 and d.stats_ws_floats >= 2 * d.cout and _addr(d.stats_ws)
            ld = lambda p: torch.from_numpy(flat(p, d.cout, np.float32).copy())
            gamma, beta, mu, istd = ld(d.bwd_gamma), ld(d.bwd_beta), ld(d.bwd_mean), ld(d.bwd_invstd)
            dy = torch.from_numpy(out.astype(np.float32))
            z = torch.from_numpy(pitched(d.bwd_z, d.n * d.ho * d.wo, d.cout, d.bwd_ldz, npdt).astype(np.float32))
            xh = (z - mu) * istd
            g = dy * self._act_grad(gamma * xh + beta, d.bwd_act, d.bwd_slope)
            row = flat(d.stats_ws, 2 * d.cout, np.float32)
            row[:d.cout] = g.sum(0).numpy()
            row[d.cout:] = (g * xh).sum(0).numpy()
            return 0
        if _addr(d.stats_ws):   # fused BatchNorm statistics of the stored values: the emulator emits a single partial row
            assert d.ups == 1 and not _addr(d.res) and d.stats_ws_floats >= 2 * d.cout
            q = torch.from_numpy(out.astype(np.float32))
            row = flat(d.stats_ws, 2 * d.cout, np.float32)
            row[:d.cout] = q.sum(0).numpy()
            row[d.cout:] = (q * q).sum(0).numpy()
        return 0

    def yh_conv2d_stem_fwd(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        x = torch.from_numpy(flat(d.x, d.n * d.cin * d.h * d.w_in, np.float32).copy()).view(d.n, d.cin, d.h, d.w_in)
        wimg = torch.from_numpy(flat(d.w, d.kh * d.kw * d.cin * d.cout_pad, np.float32).copy()).view(d.kh, d.kw, d.cin, d.cout_pad)
        w = wimg[..., :d.cout].permute(3, 2, 0, 1).contiguous()
        b = torch.from_numpy(flat(d.bias, d.cout_pad, np.float32)[:d.cout].copy())
        y = _act(F.conv2d(x, w, b, stride=d.stride, padding=d.pad), d.act, d.slope)
        assert y.shape[2] == d.ho and y.shape[3] == d.wo and d.cout % 8 == 0
        if d.dtype == hiplib.YH_I8:
            y = rnd_away(y / d.out_scale).clamp(-128, 127)
        out = pitched(d.y, d.n * d.ho * d.wo, d.cout, d.ldy, npdt)
        out[:] = y.permute(0, 2, 3, 1).reshape(-1, d.cout).numpy().astype(npdt)
        if _addr(d.stats_ws):   # the statistics epilogue of the MFMA stem: the emulator emits the sums in row 0 of its two rows
            rows = self.yh_conv2d_stem_stats_rows(d)
            if not rows:
                return -3       # YH_EUNSUPPORTED
            assert d.stats_ws_floats >= rows * 2 * d.cout
            q = torch.from_numpy(out.astype(np.float32))
            ws = flat(d.stats_ws, rows * 2 * d.cout, np.float32)
            ws[:] = 0.0
            ws[:d.cout] = q.sum(0).numpy()
            ws[d.cout:2 * d.cout] = (q * q).sum(0).numpy()
        return 0

    def yh_conv2d_stem_stats_rows(self, dref):
        d = dref._obj if hasattr(dref, '_obj') else dref
        ok = d.cin == 3 and d.kh == 3 and d.kw == 3 and d.pad == 1 and d.stride in (1, 2) and d.cout % 8 == 0 and d.dtype != hiplib.YH_I8
        return 2 if ok else 0

    def yh_maxpool2d_fwd(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        x = torch.from_numpy(pitched(d.x, d.n * d.h * d.w_in, d.c, d.ldx, npdt).astype(np.float32))
        x = x.view(d.n, d.h, d.w_in, d.c).permute(0, 3, 1, 2)
        if d.edge_zero:
            need_h = (d.ho - 1) * d.stride + d.k - d.h
            need_w = (d.wo - 1) * d.stride + d.k - d.w_in
            y = F.max_pool2d(F.pad(x, (0, max(need_w, 0), 0, max(need_h, 0)), value=0.0), d.k, d.stride, 0)
        else:
            y = F.max_pool2d(x, d.k, d.stride, d.pad_lo)
        assert y.shape[2] == d.ho and y.shape[3] == d.wo, (y.shape, d.ho, d.wo)
        out = pitched(d.y, d.n * d.ho * d.wo, d.c, d.ldy, npdt)
        out[:] = y.permute(0, 2, 3, 1).reshape(-1, d.c).numpy().astype(npdt)
        return 0

    def yh_copy_channels(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        x = pitched(d.x, d.n * d.h * d.w_in, d.c, d.ldx, npdt).reshape(d.n, d.h, d.w_in, d.c)
        if d.ups == 2:
            x = x.repeat(2, axis=1).repeat(2, axis=2)
        out = pitched(d.y, d.n * d.h * d.w_in * d.ups * d.ups, d.c, d.ldy, npdt)
        out[:] = x.reshape(-1, d.c)
        return 0

    def yh_add_channels(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        if d.amap or d.bmap:
            assert d.amap and d.bmap
            am, bm = flat(d.amap, d.c, np.int32), flat(d.bmap, d.c, np.int32)
            a = pitched(d.a, d.pixels, max(int(am.max()), 0) + 1, d.lda, npdt).astype(np.float32)
            b = pitched(d.b, d.pixels, max(int(bm.max()), 0) + 1, d.ldb, npdt).astype(np.float32)
            va = np.where(am[None, :] >= 0, a[:, np.maximum(am, 0)], 0.0)
            vb = np.where(bm[None, :] >= 0, b[:, np.maximum(bm, 0)], 0.0)
            pitched(d.y, d.pixels, d.c, d.ldy, npdt)[:] = (va + vb).astype(npdt)
            return 0
        a = pitched(d.a, d.pixels, d.c, d.lda, npdt).astype(np.float32)
        b = pitched(d.b, d.pixels, d.c, d.ldb, npdt).astype(np.float32)
        pitched(d.y, d.pixels, d.c, d.ldy, npdt)[:] = (a + b).astype(npdt)
        return 0

    def yh_yolo_decode(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        c = d.na * d.no
        p = torch.from_numpy(pitched(d.p, d.n * d.ny * d.nx, c, d.ldp, np.float32).copy())
        raw = p.view(d.n, d.ny, d.nx, d.na, d.no).permute(0, 3, 1, 2, 4).contiguous()
        io = raw.clone()
        gx = torch.arange(d.nx, dtype=torch.float32).view(1, 1, 1, d.nx)
        gy = torch.arange(d.ny, dtype=torch.float32).view(1, 1, d.ny, 1)
        aw = torch.tensor(list(d.anchor_w)[:d.na]).view(1, d.na, 1, 1)
        ah = torch.tensor(list(d.anchor_h)[:d.na]).view(1, d.na, 1, 1)
        io[..., 0] = (torch.sigmoid(raw[..., 0]) + gx) * d.stride
        io[..., 1] = (torch.sigmoid(raw[..., 1]) + gy) * d.stride
        io[..., 2] = torch.exp(raw[..., 2]) * aw * d.stride
        io[..., 3] = torch.exp(raw[..., 3]) * ah * d.stride
        io[..., 4:] = torch.sigmoid(raw[..., 4:])
        rows = d.na * d.ny * d.nx
        out = flat(d.io, d.n * d.rows_total * d.no, np.float32).reshape(d.n, d.rows_total, d.no)
        out[:, d.row_off:d.row_off + rows] = io.reshape(d.n, rows, d.no).numpy()
        if _addr(d.raw):
            flat(d.raw, raw.numel(), np.float32)[:] = raw.reshape(-1).numpy()
        return 0

    # ---- non-maximum suppression (csrc/nms.hip): records of 8 floats, int32 counts, fp32 arithmetic in the kernels' operation order
    _MIN_WH, _MAX_WH, _SEG_MAX = np.float32(2.0), np.float32(4096.0), 2048

    def _emit(self, cand, count, cap, img, recs):
        """Append records (list of 8-float rows) of image `img` at the cursor; counts run on past the bound like the kernels'."""
        cnt = flat(count, img + 1, np.int32)
        if _addr(cand):
            room = max(0, min(len(recs), cap - int(cnt[img])))
            if room:
                dst = flat(cand, (img + 1) * cap * 8, np.float32).reshape(img + 1, cap, 8)
                dst[img, int(cnt[img]):int(cnt[img]) + room] = np.asarray(recs[:room], np.float32)
        cnt[img] += len(recs)

    def _rows_to_records(self, x, row0, nc, conf, ml, mask):
        """x: (rows, 5 + nc) decoded rows of one image -> records in the kernels' filter order (utils.py:799-827)."""
        f = np.float32
        conf = f(conf)
        recs = []
        live = np.nonzero(x[:, 4] > conf)[0]
        for r in live:
            cx, cy, w, h, obj = x[r, :5]
            if not (w > self._MIN_WH and w < self._MAX_WH and h > self._MIN_WH and h < self._MAX_WH):
                continue
            box = [f(cx - w / f(2)), f(cy - h / f(2)), f(cx + w / f(2)), f(cy + h / f(2))]
            sc = (x[r, 5:] * obj).astype(np.float32)
            if ml:
                pick = [c for c in range(nc) if sc[c] > conf]
            else:
                pick = [int(np.argmax(sc))]      # first maximum, like the kernel's strict '>' scan
            for c in pick:
                if mask is not None and not mask[c]:
                    continue
                if not np.isfinite(box + [sc[c]]).all():
                    continue
                key = np.array([(row0 + r) * nc + c], np.int32).view(np.float32)[0]
                recs.append(box + [sc[c], f(c), key, f(0)])
        return recs

    def yh_nms_candidates(self, pred, n, rows, nc, conf, ml, class_mask, cand, count, cap, stream):
        x = flat(pred, n * rows * (5 + nc), np.float32).reshape(n, rows, 5 + nc)
        mask = flat(class_mask, nc, np.uint8) if _addr(class_mask) else None
        for i in range(n):
            self._emit(cand, count, cap, i, self._rows_to_records(x[i], 0, nc, conf, ml, mask))
        self.calls.append('nms_candidates')
        return 0

    def yh_yolo_decode_candidates(self, dref, conf, ml, class_mask, cand, count, cap, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        rows, nc = d.na * d.ny * d.nx, d.no - 5
        io = np.zeros((d.n, d.rows_total, d.no), np.float32)      # the decode emulation's values for this head, then the same filter
        e = DecodeDesc.from_buffer_copy(d)
        e.io, e.raw = io.ctypes.data, None
        self.yh_yolo_decode(e, stream)
        mask = flat(class_mask, nc, np.uint8) if _addr(class_mask) else None
        for i in range(d.n):
            self._emit(cand, count, cap, i, self._rows_to_records(io[i, d.row_off:d.row_off + rows], d.row_off, nc, conf, ml, mask))
        self.calls.append('decode_candidates')
        return 0

    def _order(self, rec):
        key = rec[:, 6].copy().view(np.int32)
        return np.lexsort((key, -rec[:, 4].astype(np.float64)))      # score descending, ties by ascending key

    def yh_nms_sort(self, cand, count, n, cap, mmax, sorted_, stream, cls8=None):
        src = flat(cand, n * cap * 8, np.float32).reshape(n, cap, 8)
        dst = flat(sorted_, n * cap * 8, np.float32).reshape(n, cap, 8)
        cnt = flat(count, n, np.int32)
        for i in range(n):
            m = min(int(cnt[i]), cap)
            dst[i, :m] = src[i, :m][self._order(src[i, :m])]
            if _addr(cls8):
                flat(cls8, n * cap, np.uint8).reshape(n, cap)[i, :m] = dst[i, :m, 5].astype(np.uint8)
        return 0

    def yh_nms_sort_cls(self, cand, count, n, cap, mmax, sorted_, cls8, stream):
        return self.yh_nms_sort(cand, count, n, cap, mmax, sorted_, stream, cls8=cls8)

    def yh_nms_sort_tiles(self, cand, count, n, cap, mmax, sorted_, cls8, ws, ws_bytes, stream):
        assert cap >= 256 and cap & (cap - 1) == 0 and ws_bytes >= n * cap * 12 and _addr(ws)
        return self.yh_nms_sort(cand, count, n, cap, mmax, sorted_, stream, cls8=cls8)

    @staticmethod
    def _iou_matrix(a, b):
        """IoU of every box of a against every box of b, fp32, the kernels' operation order (iou_off)."""
        f = np.float32
        iw = np.maximum(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]), f(0))
        ih = np.maximum(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]), f(0))
        inter = (iw * ih).astype(f)
        aa = ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])).astype(f)
        ab = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(f)
        with np.errstate(all='ignore'):
            return (inter / ((aa[:, None] + ab[None, :]).astype(f) - inter)).astype(f)

    def _offset(self, rec, agnostic):
        off = (rec[:, 5] * (np.float32(0) if agnostic else self._MAX_WH)).astype(np.float32)
        return (rec[:, :4] + off[:, None]).astype(np.float32)

    @staticmethod
    def _greedy(over):
        """over[i, j] (j > i): i suppresses j.  Kept indices in order."""
        m = over.shape[0]
        dead = np.zeros(m, bool)
        keep = []
        for i in range(m):
            if dead[i]:
                continue
            keep.append(i)
            dead[i + 1:] |= over[i, i + 1:]
        return keep

    def yh_nms_mask(self, sorted_, count, n, cap, mmax, thr, agnostic, mask, stream):
        rec = flat(sorted_, n * cap * 8, np.float32).reshape(n, cap, 8)
        cnt = flat(count, n, np.int32)
        words = (mmax + 63) // 64
        bits = flat(mask, n * mmax * words, np.uint64).reshape(n, mmax, words)
        for i in range(n):
            m = min(int(cnt[i]), cap)
            box = self._offset(rec[i, :m], agnostic)
            over = np.triu(self._iou_matrix(box, box) > np.float32(thr), 1)
            pad = np.zeros((m, words * 64), bool)
            pad[:, :m] = over
            bits[i, :m] = np.packbits(pad.reshape(m, words, 64), axis=2, bitorder='little').view(np.uint64).reshape(m, words)
        self.calls.append('nms_mask')
        return 0

    def yh_nms_reduce(self, mask, count, n, cap, mmax, keep_idx, n_keep, stream):
        cnt = flat(count, n, np.int32)
        words = (mmax + 63) // 64
        bits = flat(mask, n * mmax * words, np.uint64).reshape(n, mmax, words)
        kidx = flat(keep_idx, n * cap, np.int32).reshape(n, cap)
        nk = flat(n_keep, n, np.int32)
        for i in range(n):
            m = min(int(cnt[i]), cap)
            over = np.unpackbits(bits[i, :m].view(np.uint8).reshape(m, words * 8), axis=1, bitorder='little')[:, :m].astype(bool)
            keep = self._greedy(over)
            kidx[i, :len(keep)] = keep
            nk[i] = len(keep)
        return 0

    def yh_nms_class_scan(self, sorted_, cls8, count, n, cap, nc, thr, keep8, state, keep_idx, n_keep, stream):
        if nc > 255 or n > 65535:
            return hiplib.YH_EUNSUPPORTED if hasattr(hiplib, 'YH_EUNSUPPORTED') else -6
        rec = flat(sorted_, n * cap * 8, np.float32).reshape(n, cap, 8)
        cl = flat(cls8, n * cap, np.uint8).reshape(n, cap)
        cnt = flat(count, n, np.int32)
        st = flat(state, n * 8, np.uint32).reshape(n, 8)
        kidx = flat(keep_idx, n * cap, np.int32).reshape(n, cap)
        nk = flat(n_keep, n, np.int32)
        for i in range(n):
            assert st[i, 0] == 0 and st[i, 1] == 0xFFFFFFFF and st[i, 2] == 0xFFFFFFFF and st[i, 3] == 0 and st[i, 4] == 0, 'state not initialised'
            m = min(int(cnt[i]), cap)
            nk[i], st[i, 5] = 0, 0
            if m == 0:
                continue
            r = rec[i, :m]
            general = bool(np.bincount(cl[i, :m], minlength=nc).max() > self._SEG_MAX)
            ex = float(r[:, 2].max()) - float(r[:, 0].min())
            ey = float(r[:, 3].max()) - float(r[:, 1].min())
            general = general or not (ex <= 4096.0 or ey <= 4096.0)
            if general:
                st[i, 5] = 1
                continue
            box = self._offset(r, 0)
            keep = np.zeros(m, bool)
            for c in range(nc):
                pos = np.nonzero(cl[i, :m] == c)[0]
                if len(pos):
                    over = np.triu(self._iou_matrix(box[pos], box[pos]) > np.float32(thr), 1)
                    keep[pos[self._greedy(over)]] = True
            k = np.nonzero(keep)[0]
            kidx[i, :len(k)] = k
            nk[i] = len(k)
        self.calls.append('nms_class_scan')
        return 0

    def yh_nms_merge(self, sorted_, count, keep_idx, n_keep, n, cap, kmax, thr, agnostic, merge_lo, merge_hi, out, stream):
        rec = flat(sorted_, n * cap * 8, np.float32).reshape(n, cap, 8)
        cnt = flat(count, n, np.int32)
        kidx = flat(keep_idx, n * cap, np.int32).reshape(n, cap)
        nk = flat(n_keep, n, np.int32)
        res = flat(out, n * cap * 6, np.float32).reshape(n, cap, 6)
        for i in range(n):
            m, k = min(int(cnt[i]), cap), int(nk[i])
            if k == 0:
                continue
            r = rec[i, :m]
            keep = kidx[i, :k]
            boxes = r[keep, :4].copy()
            if merge_lo < m < merge_hi:
                off = self._offset(r, agnostic)
                w = (self._iou_matrix(off[keep], off) > np.float32(thr)).astype(np.float32) * r[None, :, 4]
                boxes = ((w @ r[:, :4]) / w.sum(1, keepdims=True)).astype(np.float32)
            res[i, :k, :4] = boxes
            res[i, :k, 4] = r[keep, 4]
            res[i, :k, 5] = r[keep, 5]
        return 0

    # ---- training path
    @staticmethod
    def _act_grad(u, act, slope):
        if act == 1:
            return torch.where(u > 0, torch.ones_like(u), torch.full_like(u, slope))
        if act == 2:
            return (u > 0).float()
        if act == 3:
            return ((u > 0) & (u < 6)).float()
        if act == 4:
            return torch.where(u <= -3, torch.zeros_like(u), torch.where(u >= 3, torch.ones_like(u), (2 * u + 3) / 6))
        if act == 5:
            t = torch.tanh(F.softplus(u))
            return t + u * torch.sigmoid(u) * (1 - t * t)
        return torch.ones_like(u)

    def _bn_rows(self, d, p, ld):
        return torch.from_numpy(pitched(p, d.pixels, d.c, ld, _NP[d.dtype]).astype(np.float32))

    def yh_bn_stats(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        z = self._bn_rows(d, d.z, d.ldz)
        flat(d.sum, d.c, np.float32)[:] += z.sum(0).numpy()
        flat(d.sumsq, d.c, np.float32)[:] += (z * z).sum(0).numpy()
        return 0

    def yh_conv2d_stats_rows(self, dref):
        return 1

    def yh_conv2d_bwd_stats_rows(self, dref):
        # the emulator carries a block's backward sums on every 1x1 / stride-1 data gradient (the library: only on its persistent
        # fp16 kernel, >= 262 144 pixels), so the CPU tier runs the fused plans at the sizes it can afford
        d = dref._obj if hasattr(dref, '_obj') else dref
        ok = (_addr(d.bwd_z) and d.kh == 1 and d.kw == 1 and d.stride == 1 and d.pad == 0 and d.ups == 1 and not d.out_f32
              and d.dtype in (hiplib.YH_F16, hiplib.YH_F32) and d.act == 0 and os.environ.get('YH_NO_BWD_SUMS') is None)
        return 1 if ok else 0

    # ---- COS-PTQ calibration services (csrc/calib.hip): fp32 element arithmetic, float64 sums
    def yh_ptq_search_workspace(self, count):
        return 8 * 33

    def yh_ptq_cos_search(self, t, count, scale0, n, lo, hi, do_clamp, ws, ws_bytes, cos_out, best, stream):
        v = torch.from_numpy(flat(t, int(count), np.float32).copy())
        out = flat(cos_out, int(n), np.float64)
        nt = float(v.double().pow(2).sum().sqrt())
        top, arg = -1.0, 0
        for j in range(int(n)):
            s = torch.tensor(np.float32(scale0) * np.float32(2.0 ** j))
            r = rnd_away(v / s)
            if do_clamp:
                r = r.clamp(lo, hi)
            q = (r * s).double()
            nq = float(q.pow(2).sum().sqrt())
            c = float((v.double() * q).sum()) / (nt * nq) if nt > 0 and nq > 0 else 0.0
            out[j] = c
            if c > top:
                top, arg = c, j
        flat(best, 1, np.int32)[0] = arg
        return 0

    # ---------------------------------------------------------------- input pipeline (csrc/preprocess.hip, csrc/augment.hip)
    def yh_letterbox_fwd(self, dref, stream):
        """Every arithmetic of the kernel, formula for formula from the descriptor's tables (NOT via oracle/cv2_restated.py, which
        computes the same images from the image sizes alone)."""
        d = dref._obj if hasattr(dref, '_obj') else dref
        c, nh, nw = d.c, d.new_h, d.new_w
        src = np.lib.stride_tricks.as_strided(flat(d.src, (d.h0 - 1) * d.src_pitch + d.w0 * c, np.uint8), shape=(d.h0, d.w0, c),
                                              strides=(d.src_pitch, c, 1)).astype(np.int64)
        if d.arith == 0:
            hb, vb = flat(d.hbounds, nw * 2, np.int32).reshape(nw, 2), flat(d.vbounds, nh * 2, np.int32).reshape(nh, 2)
            hk, vk = flat(d.hk, nw * d.hksize, np.int32).reshape(nw, -1).astype(np.int64), flat(d.vk, nh * d.vksize, np.int32).reshape(nh, -1).astype(np.int64)
            tmp = np.empty((d.h0, nw, c), dtype=np.int64)
            for x in range(nw):
                x0, n = hb[x]
                tmp[:, x] = np.clip(((1 << 21) + (src[:, x0:x0 + n] * hk[x, :n][None, :, None]).sum(1)) >> 22, 0, 255)
            img = np.empty((nh, nw, c), dtype=np.int64)
            for y in range(nh):
                y0, n = vb[y]
                img[y] = np.clip(((1 << 21) + (tmp[y0:y0 + n] * vk[y, :n][:, None, None]).sum(0)) >> 22, 0, 255)
        elif d.arith == 1:
            assert d.hksize == 2 and d.vksize == 2
            hb, vb = flat(d.hbounds, nw * 2, np.int32).reshape(nw, 2), flat(d.vbounds, nh * 2, np.int32).reshape(nh, 2)
            hk, vk = flat(d.hk, nw * 2, np.int32).reshape(nw, 2).astype(np.int64), flat(d.vk, nh * 2, np.int32).reshape(nh, 2).astype(np.int64)
            rows = src[:, hb[:, 0]] * hk[:, 0][None, :, None] + src[:, hb[:, 1]] * hk[:, 1][None, :, None]
            r0, r1 = rows[vb[:, 0]], rows[vb[:, 1]]
            img = np.clip((((vk[:, 0][:, None, None] * (r0 >> 4)) >> 16) + ((vk[:, 1][:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2, 0, 255)
        elif d.arith == 2:
            hb, vb = flat(d.hbounds, nw * 2, np.int32).reshape(nw, 2), flat(d.vbounds, nh * 2, np.int32).reshape(nh, 2)
            hk, vk = flat(d.hk, nw * d.hksize, np.float32).reshape(nw, -1), flat(d.vk, nh * d.vksize, np.float32).reshape(nh, -1)
            f = src.astype(np.float32)
            buf = np.zeros((d.h0, nw, c), dtype=np.float32)
            for x in range(nw):
                x0, n = hb[x]
                for u in range(n):
                    buf[:, x] = buf[:, x] + f[:, x0 + u] * hk[x, u]
            img = np.empty((nh, nw, c), dtype=np.int64)
            for y in range(nh):
                y0, n = vb[y]
                acc = vk[y, 0] * buf[y0]
                for t in range(1, n):
                    acc = acc + vk[y, t] * buf[y0 + t]
                img[y] = np.clip(np.rint(acc), 0, 255)
        else:
            fx, fy = d.hksize, d.vksize
            img = np.zeros((nh, nw, c), dtype=np.int64)
            for y in range(nh):
                for x in range(nw):
                    cell = src[y * fy:min(y * fy + fy, d.h0), x * fx:min(x * fx + fx, d.w0)]
                    if cell.size == 0:
                        continue
                    sm = cell.sum((0, 1))
                    if y * fy + fy <= d.h0 and x < d.w0 // fx:
                        img[y, x] = (sm + 2) >> 2 if (fx, fy) == (2, 2) else np.clip(np.rint(sm.astype(np.float32) * (np.float32(1) / np.float32(fx * fy))), 0, 255)
                    else:
                        img[y, x] = np.clip(np.rint(sm.astype(np.float32) / np.float32(cell.shape[0] * cell.shape[1])), 0, 255)
        full = np.full((d.out_h, d.out_w, c), d.pad_value, dtype=np.int64)
        full[d.top:d.top + nh, d.left:d.left + nw] = img
        if d.out_u8:
            flat(d.dst, d.out_h * d.out_w * c, np.uint8)[:] = full.astype(np.uint8).reshape(-1)
        else:
            chw = full.transpose(2, 0, 1)
            if d.swap_rb and c == 3:
                chw = chw[::-1]
            flat(d.dst, c * d.out_h * d.out_w, np.float32)[:] = (chw.astype(np.float32) * np.float32(d.scale) + np.float32(d.shift)).reshape(-1)
        return 0

    def yh_mosaic_affine_hsv(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        if d.arith != 1:
            raise NotImplementedError('the Pillow arithmetic of this kernel is restated by engine.preprocess.mosaic_reference')
        c, H, W = d.c, d.canvas_h, d.canvas_w
        canvas = np.full((H, W, c), d.pad_value, dtype=np.int64)
        for k in range(4):
            if d.x2a[k] > d.x1a[k] and d.y2a[k] > d.y1a[k]:
                img = np.lib.stride_tricks.as_strided(flat(d.src[k], (d.src_h[k] - 1) * d.src_pitch[k] + d.src_w[k] * c, np.uint8),
                                                      shape=(d.src_h[k], d.src_w[k], c), strides=(d.src_pitch[k], c, 1))
                canvas[d.y1a[k]:d.y2a[k], d.x1a[k]:d.x2a[k]] = img[d.y1b[k]:d.y1b[k] + d.y2a[k] - d.y1a[k], d.x1b[k]:d.x1b[k] + d.x2a[k] - d.x1a[k]]
        inv = [float(v) for v in d.inv]
        X = np.arange(d.out_w)
        xo = (d.out_w - 1 - X) if d.flip_lr else X
        Y = np.arange(d.out_h)
        rint = lambda v: np.rint(v).astype(np.int64)
        ad, bd = rint(inv[0] * xo.astype(np.float64) * 1024.0), rint(inv[3] * xo.astype(np.float64) * 1024.0)
        X0, Y0 = rint((inv[1] * Y.astype(np.float64) + inv[2]) * 1024.0) + 16, rint((inv[4] * Y.astype(np.float64) + inv[5]) * 1024.0) + 16
        XX, YY = (X0[:, None] + ad[None, :]) >> 5, (Y0[:, None] + bd[None, :]) >> 5
        sx, sy, fx, fy = np.clip(XX >> 5, -32768, 32767), np.clip(YY >> 5, -32768, 32767), XX & 31, YY & 31

        def px(yy, xx):
            ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
            return np.where(ok[..., None], canvas[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], d.pad_value)
        acc = (px(sy, sx) * ((32 - fy) * (32 - fx) * 32)[..., None] + px(sy, sx + 1) * ((32 - fy) * fx * 32)[..., None] +
               px(sy + 1, sx) * (fy * (32 - fx) * 32)[..., None] + px(sy + 1, sx + 1) * (fy * fx * 32)[..., None])
        img = np.clip((acc + (1 << 14)) >> 15, 0, 255)
        if d.hsv and c == 3:
            lut = flat(d.lut, 768, np.uint8).reshape(3, 256).astype(np.int64)
            r, g, b = img[..., 0], img[..., 1], img[..., 2]
            v, vmin = np.maximum(np.maximum(b, g), r), np.minimum(np.minimum(b, g), r)
            diff = v - vmin
            sdiv = np.where(v > 0, rint(float(255 << 12) / np.maximum(v, 1).astype(np.float64)), 0)
            hdiv = np.where(diff > 0, rint(float(180 << 12) / (6.0 * np.maximum(diff, 1).astype(np.float64))), 0)
            s = (diff * sdiv + (1 << 11)) >> 12
            h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
            h = (h * hdiv + (1 << 11)) >> 12
            h = np.where(h < 0, h + 180, h)
            Hh, Ss, Vv = lut[0][np.clip(h, 0, 255)], lut[1][s], lut[2][v]
            f32 = np.float32
            fs, fv = Ss.astype(f32) * f32(1.0 / 255.0), Vv.astype(f32) * f32(1.0 / 255.0)
            fh = np.fmod(Hh.astype(f32) * (f32(6.0) / f32(180.0)), f32(6.0)).astype(f32)
            sector = np.floor(fh).astype(np.int64)
            fh = (fh - sector.astype(f32)).astype(f32)
            bad = (sector < 0) | (sector > 5)
            sector, fh = np.where(bad, 0, sector), np.where(bad, f32(0), fh).astype(f32)
            one = f32(1)
            tab = np.stack([fv, fv * (one - fs), fv * (one - fs * fh), fv * (one - fs * (one - fh))], -1).astype(f32)
            sel = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])[sector]      # (b, g, r)
            bgr = np.take_along_axis(tab, sel, -1)
            bgr = np.where((Ss == 0)[..., None], fv[..., None], bgr).astype(f32)
            img = np.clip(np.rint(bgr * f32(255.0)), 0, 255).astype(np.int64)[..., ::-1]
        chw = img.transpose(2, 0, 1)
        n = c * d.out_h * d.out_w
        if d.out_dtype == 0:
            flat(d.dst, n, np.uint8)[:] = chw.astype(np.uint8).reshape(-1)
        else:
            dt = np.float32 if d.out_dtype == 1 else np.float16
            flat(d.dst, n, dt)[:] = (chw.astype(np.float32) / np.float32(d.divisor)).astype(dt).reshape(-1)
        return 0

    def yh_absmax(self, t, count, ws, ws_bytes, out, stream):
        flat(out, 1, np.float32)[0] = np.abs(flat(t, int(count), np.float32)).max()
        return 0

    def yh_bn_finalize(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        if d.nparts > 0:
            part = flat(d.ws, d.nparts * 2 * d.c, np.float32).reshape(d.nparts, 2, d.c)
            flat(d.sum, d.c, np.float32)[:] += part[:, 0].sum(0)
            flat(d.sumsq, d.c, np.float32)[:] += part[:, 1].sum(0)
        P = float(d.pixels)
        mean = flat(d.sum, d.c, np.float32) / np.float32(P)
        var = np.maximum(flat(d.sumsq, d.c, np.float32) / np.float32(P) - mean * mean, 0).astype(np.float32)
        flat(d.mean, d.c, np.float32)[:] = mean
        flat(d.invstd, d.c, np.float32)[:] = 1.0 / np.sqrt(var + np.float32(d.eps))
        if _addr(d.running_mean):
            rm = flat(d.running_mean, d.c, np.float32)
            rm[:] = (1 - d.momentum) * rm + d.momentum * mean
        if _addr(d.running_var):
            rv = flat(d.running_var, d.c, np.float32)
            rv[:] = (1 - d.momentum) * rv + d.momentum * var * (P / (P - 1) if P > 1 else 1.0)
        return 0

    def _bn_u(self, d, z):
        """pre-activation u and xhat from the stored conv output z ([pixels, c] fp32)."""
        if _addr(d.gamma):
            g = torch.from_numpy(flat(d.gamma, d.c, np.float32).copy())
            b = torch.from_numpy(flat(d.beta, d.c, np.float32).copy())
            mu = torch.from_numpy(flat(d.mean, d.c, np.float32).copy())
            istd = torch.from_numpy(flat(d.invstd, d.c, np.float32).copy())
            xh = (z - mu) * istd
            return g * xh + b, xh, g, istd
        b = torch.from_numpy(flat(d.beta, d.c, np.float32).copy()) if _addr(d.beta) else torch.zeros(d.c)
        return z + b, z, None, None

    def yh_bn_act_fwd(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        u, _, _, _ = self._bn_u(d, self._bn_rows(d, d.z, d.ldz))
        y = _act(u, d.act, d.slope)
        if _addr(d.res):
            y = y + self._bn_rows(d, d.res, d.ldr)
        if d.ups == 2:
            assert d.n * d.h * d.w_in == d.pixels
            y = y.view(d.n, d.h, d.w_in, d.c).repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(-1, d.c)
        pitched(d.out, y.shape[0], d.c, d.ldo, npdt)[:] = y.numpy().astype(npdt)
        return 0

    def yh_bn_act_bwd_reduce(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        if d.nparts > 0:        # rows of [sum g | sum g xhat] left by the data gradient that completed dy (yh_conv_desc.bwd_z)
            assert d.ws_floats >= d.nparts * 2 * d.c
            rows = flat(d.ws, d.nparts * 2 * d.c, np.float32).reshape(d.nparts, 2, d.c)
            flat(d.sum, d.c, np.float32)[:] += rows[:, 0].sum(0)
            flat(d.sumsq, d.c, np.float32)[:] += rows[:, 1].sum(0)
            return 0
        u, xh, _, _ = self._bn_u(d, self._bn_rows(d, d.z, d.ldz))
        g = self._bn_rows(d, d.dy, d.lddy) * self._act_grad(u, d.act, d.slope)
        flat(d.sum, d.c, np.float32)[:] += g.sum(0).numpy()
        flat(d.sumsq, d.c, np.float32)[:] += (g * xh).sum(0).numpy()
        return 0

    # ---- first block, whole backward in one pass (csrc/stem_bwd.hip): the kernel's sums, operand roundings included
    @staticmethod
    def _stem_bwd_ok(d):
        return d.n > 0 and d.h > 0 and d.w_in > 0 and d.cin in (1, 3) and d.cout in (16, 32)

    def yh_stem_bwd_workspace(self, dref):
        d = dref._obj if hasattr(dref, '_obj') else dref
        return 4 if self._stem_bwd_ok(d) else 0

    def yh_stem_bwd(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        if not self._stem_bwd_ok(d):
            return -3       # YH_EUNSUPPORTED
        P, c = d.n * d.h * d.w_in, d.cout
        ld = lambda p: torch.from_numpy(flat(p, c, np.float32).copy())
        gamma, beta, mu, istd = ld(d.gamma), ld(d.beta), ld(d.mean), ld(d.invstd)
        z = torch.from_numpy(pitched(d.z, P, c, d.ldz, np.float16).astype(np.float32))
        if _addr(d.dz1):
            # the next conv's data gradient fused in: dy = conv_transpose(dz1, W1) (3x3 / s2 / p1), rounded to f16 as the launch
            # of its own would have stored it; W1[k][c][r][s] = image[c][2 - r][2 - s][k] (yh_conv_pack_weights_dgrad layout)
            assert c == 32 and d.k1 == 64 and d.h1 == (d.h - 1) // 2 + 1 and d.w1_in == (d.w_in - 1) // 2 + 1
            img = torch.from_numpy(flat(d.w1, c * 9 * d.k1_pad, np.float16).astype(np.float32)).view(c, 3, 3, d.k1_pad)[..., :d.k1]
            w1 = img.flip(1, 2).permute(3, 0, 1, 2).contiguous()                      # [k][c][r][s]
            dz1 = torch.from_numpy(pitched(d.dz1, d.n * d.h1 * d.w1_in, d.k1, d.lddz1, np.float16).astype(np.float32))
            dz1 = dz1.view(d.n, d.h1, d.w1_in, d.k1).permute(0, 3, 1, 2)
            op = (d.h - (2 * (d.h1 - 1) + 1), d.w_in - (2 * (d.w1_in - 1) + 1))
            dy = F.conv_transpose2d(dz1, w1, stride=2, padding=1, output_padding=op).permute(0, 2, 3, 1).reshape(P, c).half().float()
        else:
            dy = torch.from_numpy(pitched(d.dy, P, c, d.lddy, np.float16).astype(np.float32))
        xh = (z - mu) * istd
        g = dy * self._act_grad(gamma * xh + beta, d.act, d.slope)
        s1, s2 = g.double().sum(0), (g * xh).double().sum(0)
        x = torch.from_numpy(flat(d.x, d.n * d.cin * d.h * d.w_in, np.float32).copy()).view(d.n, d.cin, d.h, d.w_in)
        # X[p][k], k = (r * 3 + s) * cin + ci, as f16 MFMA operands; g and xhat likewise
        cols = F.unfold(x.half().float(), 3, padding=1).view(d.n, d.cin, 9, d.h * d.w_in).permute(0, 3, 2, 1).reshape(P, 9 * d.cin).double()
        gq, xq = g.half().double(), xh.half().double()
        Q, R, SX = gq.t() @ cols, xq.t() @ cols, cols.sum(0)
        dw = (gamma * istd).double()[:, None] * (Q - (s1 / P)[:, None] * SX[None] - (s2 / P)[:, None] * R)      # [c][k]
        dw = dw.view(c, 9, d.cin).permute(0, 2, 1).reshape(-1)                                                   # -> [c][ci][r][s]
        flat(d.dw, c * d.cin * 9, np.float32)[:] += dw.float().numpy()
        flat(d.dbeta, c, np.float32)[:] += s1.float().numpy()
        flat(d.dgamma, c, np.float32)[:] += s2.float().numpy()
        return 0

    def yh_bn_act_bwd_apply(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        u, xh, gamma, istd = self._bn_u(d, self._bn_rows(d, d.z, d.ldz))
        g = self._bn_rows(d, d.dy, d.lddy) * self._act_grad(u, d.act, d.slope)
        if gamma is not None:
            s1 = torch.from_numpy(flat(d.sum, d.c, np.float32).copy())
            s2 = torch.from_numpy(flat(d.sumsq, d.c, np.float32).copy())
            g = gamma * istd * (g - s1 / d.pixels - xh * s2 / d.pixels)
        pitched(d.out, d.pixels, d.c, d.ldo, npdt)[:] = g.numpy().astype(npdt)
        return 0

    def yh_conv_pack_weights_dgrad_phase(self, dtype, w, cout, cin, kh, kw, pad, pa, pb, cout_k, m_pad, packed, kh_p, kw_p, stream):
        npdt = _NP[dtype]
        wt = torch.from_numpy(flat(w, cout * cin * kh * kw, np.float32).copy()).view(cout, cin, kh, kw)
        khp, kwp = (pa + pad) // 2 + 1, (pb + pad) // 2 + 1
        img = torch.zeros(m_pad, khp, kwp, cout_k)
        for t in range(khp):
            for u in range(kwp):
                r, s_ = pa + pad - 2 * t, pb + pad - 2 * u
                if 0 <= r < kh and 0 <= s_ < kw:
                    img[:cin, t, u, :cout] = wt[:, :, r, s_].t()
        flat(packed, img.numel(), npdt)[:] = img.reshape(-1).numpy().astype(npdt)
        for ptr, val in ((kh_p, khp), (kw_p, kwp)):
            if ptr is not None:
                (ptr._obj if hasattr(ptr, '_obj') else ptr.contents).value = val
        return 0

    # ---- depthwise / squeeze-excite backward
    def yh_dw_wgrad_workspace(self, dref):
        return 0

    def yh_dw_wgrad(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        x = torch.from_numpy(pitched(d.x, d.n * d.h * d.w_in, d.c, d.ldx, npdt).astype(np.float32)).view(d.n, d.h, d.w_in, d.c)
        dz = torch.from_numpy(pitched(d.dz, d.n * d.ho * d.wo, d.c, d.lddz, npdt).astype(np.float32)).view(d.n, d.ho, d.wo, d.c)
        gw = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2), (d.c, 1, d.k, d.k), dz.permute(0, 3, 1, 2), stride=d.stride,
                                         padding=d.pad, groups=d.c)
        flat(d.dw, gw.numel(), np.float32)[:] += gw.reshape(-1).numpy()
        return 0

    def yh_dw_dgrad(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        dz = torch.from_numpy(pitched(d.dz, d.n * d.ho * d.wo, d.c, d.lddz, npdt).astype(np.float32)).view(d.n, d.ho, d.wo, d.c)
        w = torch.from_numpy(flat(d.w, d.k * d.k * d.c, npdt).astype(np.float32)).view(d.k, d.k, d.c).permute(2, 0, 1).unsqueeze(1)
        gx = torch.nn.grad.conv2d_input((d.n, d.c, d.h, d.w_in), w.contiguous(), dz.permute(0, 3, 1, 2), stride=d.stride,
                                        padding=d.pad, groups=d.c)
        out = pitched(d.dx, d.n * d.h * d.w_in, d.c, d.lddx, npdt)
        v = gx.permute(0, 2, 3, 1).reshape(-1, d.c).numpy()
        out[:] = (out.astype(np.float32) + v if d.accumulate else v).astype(npdt)
        return 0

    def yh_se_bwd(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        hw = d.h * d.w_in
        x = torch.from_numpy(pitched(d.x, d.n * hw, d.c, d.ldx, npdt).astype(np.float32)).view(d.n, hw, d.c)
        dy = torch.from_numpy(pitched(d.dy, d.n * hw, d.c, d.lddy, npdt).astype(np.float32)).view(d.n, hw, d.c)
        w1 = torch.from_numpy(flat(d.w1, d.cr * d.c, np.float32).copy()).view(d.cr, d.c)
        w2 = torch.from_numpy(flat(d.w2, d.c * d.cr, np.float32).copy()).view(d.c, d.cr)
        pooled = torch.from_numpy(flat(d.pooled, d.n * d.c, np.float32).copy()).view(d.n, d.c)
        gate = torch.from_numpy(flat(d.gate, d.n * d.c, np.float32).copy()).view(d.n, d.c)
        with torch.enable_grad():
            pv, w1v, w2v = pooled.clone().requires_grad_(), w1.clone().requires_grad_(), w2.clone().requires_grad_()
            g = F.relu6(F.linear(F.relu(F.linear(pv, w1v)), w2v) + 3.0) / 6.0
            dg = (dy * x).sum(1)
            g.backward(dg)
        dxv = dy * gate.unsqueeze(1) + pv.grad.unsqueeze(1) / hw
        out = pitched(d.dx, d.n * hw, d.c, d.lddx, npdt)
        v = dxv.reshape(-1, d.c).numpy()
        out[:] = (out.astype(np.float32) + v if d.accumulate else v).astype(npdt)
        flat(d.dw1, d.cr * d.c, np.float32)[:] += w1v.grad.reshape(-1).numpy()
        flat(d.dw2, d.c * d.cr, np.float32)[:] += w2v.grad.reshape(-1).numpy()
        return 0

    # ---- fused loss
    @staticmethod
    def _loss_view(ptr, d, sb, sa, sy, sx):
        """Strided numpy view (bs, na, ny, nx, no) of a raw head / its gradient at a host address."""
        n = (d.bs - 1) * sb + (d.na - 1) * sa + (d.ny - 1) * sy + (d.nx - 1) * sx + d.no
        base = flat(ptr, n, np.float32)
        return np.lib.stride_tricks.as_strided(base, shape=(d.bs, d.na, d.ny, d.nx, d.no),
                                               strides=(4 * sb, 4 * sa, 4 * sy, 4 * sx, 4), writeable=True)

    @staticmethod
    def _loss_assign(d):
        """build_targets for one head from the descriptor: (idx (nb, 4) long, tbox (nb, 4), tcls (nb), anchors (nb, 2), mask)."""
        if not d.nt:
            z = torch.zeros
            return z(0, 4, dtype=torch.long), z(0, 4), z(0, dtype=torch.long), z(0, 2), 0
        tg = torch.from_numpy(flat(d.targets, 6 * d.nt, np.float32).reshape(d.nt, 6).copy())
        anc = torch.from_numpy(flat(d.anchors, 2 * d.na, np.float32).reshape(d.na, 2).copy())
        t = tg * torch.tensor([1, 1, d.nx, d.ny, d.nx, d.ny], dtype=torch.float32)
        inter = torch.min(anc[:, None], t[None, :, 4:6]).prod(2)
        iou = inter / (anc[:, None].prod(2) + t[None, :, 4:6].prod(2) - inter)
        a = torch.arange(d.na).view(-1, 1).repeat(1, d.nt).view(-1)
        keep = iou.view(-1) > torch.tensor(d.iou_t, dtype=torch.float32)
        t, a = t.repeat(d.na, 1)[keep], a[keep]
        b, c = t[:, 0].long(), t[:, 1].long()
        gi, gj = t[:, 2].long(), t[:, 3].long()
        bad_c = (c < 0) | (c >= d.nc)
        bad_i = (b < 0) | (b >= d.bs) | (gi < 0) | (gi >= d.nx) | (gj < 0) | (gj >= d.ny)
        mask = (1 if bool(bad_c.any()) else 0) | (2 if bool(bad_i.any()) else 0)
        ok = ~(bad_c | bad_i)
        t, a, b, c, gi, gj = t[ok], a[ok], b[ok], c[ok], gi[ok], gj[ok]
        tbox = torch.cat((t[:, 2:4] - t[:, 2:4].floor(), t[:, 4:6]), 1)
        return torch.stack((b, a, gj, gi), 1), tbox, c, anc[a], mask

    def _loss_terms(self, d, p):
        """(sum (1 - giou), sum obj bce, sum cls bce, tobj, nb, mask) of one head in torch, differentiable in p."""
        tobj = torch.zeros(d.bs, d.na, d.ny, d.nx)
        lbox = p.sum() * 0
        lcls = p.sum() * 0
        idx, tbox, tcls, anc, mask = self._loss_assign(d)
        nb = idx.shape[0]
        if nb:
            b, a, gj, gi = idx.t()
            ps = p[b, a, gj, gi]
            pxy = torch.sigmoid(ps[:, 0:2])
            pwh = torch.exp(ps[:, 2:4]).clamp(max=1e3) * anc
            pb = torch.cat((pxy, pwh), 1)
            ax1, ax2 = pb[:, 0] - pb[:, 2] / 2, pb[:, 0] + pb[:, 2] / 2
            ay1, ay2 = pb[:, 1] - pb[:, 3] / 2, pb[:, 1] + pb[:, 3] / 2
            bx1, bx2 = tbox[:, 0] - tbox[:, 2] / 2, tbox[:, 0] + tbox[:, 2] / 2
            by1, by2 = tbox[:, 1] - tbox[:, 3] / 2, tbox[:, 1] + tbox[:, 3] / 2
            inter = (torch.min(ax2, bx2) - torch.max(ax1, bx1)).clamp(0) * (torch.min(ay2, by2) - torch.max(ay1, by1)).clamp(0)
            union = ((ax2 - ax1) * (ay2 - ay1) + 1e-16) + (bx2 - bx1) * (by2 - by1) - inter
            iou = inter / union
            hull = (torch.max(ax2, bx2) - torch.min(ax1, bx1)) * (torch.max(ay2, by2) - torch.min(ay1, by1)) + 1e-16
            giou = iou - (hull - union) / hull
            lbox = (1.0 - giou).sum()
            tobj[b, a, gj, gi] = (1.0 - d.gr) + d.gr * giou.detach().clamp(0)
            if d.nc > 1:
                t = torch.full_like(ps[:, 5:], d.cn)
                t[range(nb), tcls] = d.cp
                lcls = F.binary_cross_entropy_with_logits(ps[:, 5:], t, pos_weight=torch.tensor([d.cls_pw]), reduction='sum')
        lobj = F.binary_cross_entropy_with_logits(p[..., 4], tobj, pos_weight=torch.tensor([d.obj_pw]), reduction='sum')
        return lbox, lobj, lcls, tobj, nb, mask

    def yh_yolo_loss_fwd(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        p = torch.from_numpy(self._loss_view(d.p, d, d.sb, d.sa, d.sy, d.sx).copy())
        lbox, lobj, lcls, tobj, nb, mask = self._loss_terms(d, p)
        flat(d.tobj, tobj.numel(), np.float32)[:] = tobj.reshape(-1).numpy()
        sums = flat(d.sums, 3, np.float32)
        sums[0] += float(lbox)
        sums[1] += float(lobj)
        sums[2] += float(lcls)
        count = flat(d.count, 2, np.int32)
        count[0] += nb
        count[1] |= mask
        return 0

    def yh_yolo_loss_bwd(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        scale = float(flat(d.scale, 1, np.float32)[0])
        nb = max(int(flat(d.count, 2, np.int32)[0]), 1)
        with torch.enable_grad():
            p = torch.from_numpy(self._loss_view(d.p, d, d.sb, d.sa, d.sy, d.sx).copy()).requires_grad_()
            lbox, lobj, lcls, _, _, _ = self._loss_terms(d, p)
            cells = d.bs * d.na * d.ny * d.nx
            (scale * (d.g_box / nb * lbox + d.g_obj / cells * lobj + d.g_cls / (nb * max(d.nc, 1)) * lcls)).backward()
        self._loss_view(d.grad, d, d.gb, d.ga, d.gy, d.gx)[:] = p.grad.numpy()
        return 0

    def yh_pack_batch(self, items, n_items, stream):
        """The batched packer = the single-layer packers applied to every item of the (host-resident) table."""
        arr = (PackItem * n_items).from_address(_addr(items))
        for it in arr:
            if it.mode == 0:
                rc = self.yh_conv_pack_weights(it.dtype, it.w, it.bias, None, None, None, None, 0.0, None, it.cout, it.cin, it.kh,
                                               it.kw, it.k_pad, it.m_pad, it.packed, it.bias_out, stream)
            elif it.mode == 1:
                rc = self.yh_conv_pack_weights_dgrad(it.dtype, it.w, it.cout, it.cin, it.kh, it.kw, it.k_pad, it.m_pad, it.packed,
                                                     stream)
            elif it.mode == 4:
                rc = self.yh_dw_pack_weights(it.dtype, it.w, it.bias, None, None, None, None, 0.0, None, it.cout, it.kh, it.k_pad,
                                             it.packed, it.bias_out, stream)
            elif it.mode == 5:
                wt = torch.from_numpy(flat(it.w, it.cout * it.cin * it.kh * it.kw, np.float32).copy()).view(it.cout, it.cin, it.kh, it.kw)
                img = torch.zeros(it.m_pad, 2, 2, it.k_pad)
                for ph in range(4):
                    for t in range(2):
                        for u in range(2):
                            fr, fs = (ph >> 1) + it.pad - 2 * t, (ph & 1) + it.pad - 2 * u
                            if 0 <= fr < it.kh and 0 <= fs < it.kw:
                                img[ph * it.cout_pad:ph * it.cout_pad + it.cin, t, u, :it.cout] = wt[:, :, fr, fs].t()
                flat(it.packed, img.numel(), _NP[it.dtype])[:] = img.reshape(-1).numpy().astype(_NP[it.dtype])
                rc = 0
            elif it.mode == 2:
                rc = self.yh_conv_pack_weights_dgrad_phase(it.dtype, it.w, it.cout, it.cin, it.kh, it.kw, it.pad, it.pa, it.pb,
                                                           it.k_pad, it.m_pad, it.packed, None, None, stream)
            else:
                rc = self.yh_stem_pack_weights(it.w, it.bias, None, None, None, None, 0.0, it.cout, it.cin, it.kh, it.kw,
                                               it.cout_pad, it.packed, it.bias_out, stream)
            assert rc == 0
        return 0

    def yh_conv2d_wgrad_workspace(self, dref):
        return 0

    def yh_conv2d_wgrad_kernel(self, dref):
        return 1        # the emulation has one weight-gradient form

    def yh_bn_reduce_workspace(self, dref):
        d = dref._obj if hasattr(dref, '_obj') else dref
        return d.nparts * 2 * d.c if d.nparts > 0 else 0

    def yh_conv_pack_weights_dgrad(self, dtype, w, cout, cin, kh, kw, cout_k, m_pad, packed, stream):
        npdt = _NP[dtype]
        wt = torch.from_numpy(flat(w, cout * cin * kh * kw, np.float32).copy()).view(cout, cin, kh, kw)
        img = torch.zeros(m_pad, kh, kw, cout_k)
        img[:cin, :, :, :cout] = wt.flip(2, 3).permute(1, 2, 3, 0)
        flat(packed, img.numel(), npdt)[:] = img.reshape(-1).numpy().astype(npdt)
        return 0

    def yh_conv2d_wgrad(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        vec = 8 if d.dtype == hiplib.YH_F16 else 4
        assert d.ldx % vec == 0 and d.lddz % vec == 0 and _addr(d.x) % 16 == 0 and _addr(d.dz) % 16 == 0
        x = torch.from_numpy(pitched(d.x, d.n * d.h * d.w_in, d.cin, d.ldx, npdt).astype(np.float32))
        x = x.view(d.n, d.h, d.w_in, d.cin).permute(0, 3, 1, 2)
        dz = torch.from_numpy(pitched(d.dz, d.n * d.ho * d.wo, d.cout, d.lddz, npdt).astype(np.float32))
        dz = dz.view(d.n, d.ho, d.wo, d.cout).permute(0, 3, 1, 2)
        gw = torch.nn.grad.conv2d_weight(x, (d.cout, d.cin, d.kh, d.kw), dz, stride=d.stride, padding=d.pad)
        if 0 < d.cin_w < d.cin:
            gw = gw[:, :d.cin_w].contiguous()
        flat(d.dw, gw.numel(), np.float32)[:] += gw.reshape(-1).numpy()
        return 0

    def yh_nchw_to_nhwc(self, x, y, n, c, h, w, c_pad, ldy, dtype, stream):
        src = flat(x, n * c * h * w, np.float32).reshape(n, c, h, w)
        dst = pitched(y, n * h * w, c_pad, ldy, _NP[dtype])
        dst[:, :c] = src.transpose(0, 2, 3, 1).reshape(-1, c).astype(_NP[dtype])
        dst[:, c:] = 0
        return 0

    def _layout(self, d, stream):
        return self.yh_nchw_to_nhwc(d.x, d.y, d.n, d.c, d.h, d.w_in, d.c_pad, d.ldy, d.dtype, stream)

    def yh_stem_wgrad(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        assert d.cin == 3 and d.kh == 3 and d.kw == 3
        x = torch.from_numpy(flat(d.x, d.n * d.cin * d.h * d.w_in, np.float32).copy()).view(d.n, d.cin, d.h, d.w_in)
        dz = torch.from_numpy(pitched(d.dz, d.n * d.ho * d.wo, d.cout, d.lddz, npdt).astype(np.float32))
        dz = dz.view(d.n, d.ho, d.wo, d.cout).permute(0, 3, 1, 2)
        gw = torch.nn.grad.conv2d_weight(x, (d.cout, d.cin, d.kh, d.kw), dz, stride=d.stride, padding=d.pad)
        flat(d.dw, gw.numel(), np.float32)[:] += gw.reshape(-1).numpy()
        return 0

    def yh_dilate2(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        src = pitched(d.x, d.n * d.h * d.w_in, d.c, d.ldx, npdt).reshape(d.n, d.h, d.w_in, d.c)
        assert 2 * d.h - 1 <= d.big_h <= 2 * d.h and 2 * d.w_in - 1 <= d.big_w <= 2 * d.w_in
        dst = pitched(d.y, d.n * d.big_h * d.big_w, d.c, d.ldy, npdt)
        idx = (np.arange(d.n)[:, None, None] * d.big_h + 2 * np.arange(d.h)[None, :, None]) * d.big_w \
            + 2 * np.arange(d.w_in)[None, None, :]
        dst[idx.reshape(-1)] = src.reshape(-1, d.c)
        return 0

    def yh_upsample2_bwd(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        assert d.big_h == 2 * d.h and d.big_w == 2 * d.w_in
        big = torch.from_numpy(pitched(d.x, d.n * 4 * d.h * d.w_in, d.c, d.ldx, npdt).astype(np.float32))
        big = big.view(d.n, d.h, 2, d.w_in, 2, d.c)
        small = (big[:, :, 0, :, 0] + big[:, :, 0, :, 1]) + (big[:, :, 1, :, 0] + big[:, :, 1, :, 1])
        pitched(d.y, d.n * d.h * d.w_in, d.c, d.ldy, npdt)[:] = small.reshape(-1, d.c).numpy().astype(npdt)
        return 0

    def yh_maxpool2d_bwd(self, dref, stream):
        """Autograd of the emulated forward pooling (F.max_pool2d picks the first maximum in scan order)."""
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        x = torch.from_numpy(pitched(d.x, d.n * d.h * d.w_in, d.c, d.ldx, npdt).astype(np.float32))
        gy = torch.from_numpy(pitched(d.dy, d.n * d.ho * d.wo, d.c, d.lddy, npdt).astype(np.float32))
        with torch.enable_grad():   # this runs inside an autograd backward, where grad mode is off
            x = x.view(d.n, d.h, d.w_in, d.c).permute(0, 3, 1, 2).contiguous().requires_grad_()
            if d.edge_zero:
                y = F.max_pool2d(F.pad(x, (0, 1, 0, 1)), d.k, d.stride)
            else:
                y = F.max_pool2d(x, d.k, d.stride, d.pad_lo)
            assert y.shape[2] == d.ho and y.shape[3] == d.wo
            y.backward(gy.view(d.n, d.ho, d.wo, d.c).permute(0, 3, 1, 2))
        out = pitched(d.dx, d.n * d.h * d.w_in, d.c, d.lddx, npdt)
        out[:] = (out.astype(np.float32) + x.grad.permute(0, 2, 3, 1).reshape(-1, d.c).numpy()).astype(npdt)
        return 0

    def yh_cast_f32(self, dref, stream):
        d = dref._obj if hasattr(dref, '_obj') else dref
        npdt = _NP[d.dtype]
        pitched(d.y, d.pixels, d.c, d.ldy, npdt)[:] = pitched(d.x, d.pixels, d.c, d.ldx, np.float32).astype(npdt)
        return 0

    # ---- plans
    def yh_plan_create(self):
        h = self.next
        self.next += 1
        self.plans[h] = dict(ops=[], slots={})
        return h

    def yh_plan_destroy(self, h):
        self.plans.pop(_addr(h), None)

    def yh_plan_add(self, h, kind, dref, nbytes):
        d = dref._obj if hasattr(dref, '_obj') else dref
        cls = _DESC[kind]
        assert nbytes == C.sizeof(cls) and isinstance(d, cls)
        copy = cls.from_buffer_copy(bytes(d))
        self.plans[_addr(h)]['ops'].append((kind, copy, []))
        return len(self.plans[_addr(h)]['ops']) - 1

    def yh_plan_add_fixup(self, h, op, field_offset, slot, byte_offset):
        self.plans[_addr(h)]['ops'][op][2].append((field_offset, slot, byte_offset))
        return 0

    def yh_plan_bind_slot(self, h, slot, p):
        self.plans[_addr(h)]['slots'][slot] = _addr(p)      # hiplib.SLOT_NULL = 'bound to nothing': fixups then write NULL
        return 0

    def yh_plan_num_ops(self, h):
        return len(self.plans[_addr(h)]['ops'])

    # lanes: the emulation replays the LATEST schedule the declared dependencies allow - a side-lane op runs only when a
    # main-lane op that waits for it comes up, or at the join at the end of the range - so a missing write-after-read dependency
    # (a main-lane op overwriting what a side-lane op still has to read) shows up as wrong numbers on the CPU tier
    def yh_plan_set_async_reduce(self, h, enable):
        # the emulator runs every op to completion in program order: the reduce launches have nowhere to overlap
        self.plans[_addr(h)]['async_reduce'] = bool(enable)
        return 0

    def yh_plan_set_lane(self, h, op, lane):
        self.plans[_addr(h)].setdefault('lane', {})[op] = lane
        return 0

    def yh_plan_add_dep(self, h, op, dep):
        assert 0 <= dep < op
        self.plans[_addr(h)].setdefault('deps', {}).setdefault(op, []).append(dep)
        return 0

    def yh_plan_run(self, h, stream):
        return self.yh_plan_run_range(h, 0, self.yh_plan_num_ops(h), stream)

    def yh_plan_run_range(self, h, first, last, stream):
        plan = self.plans[_addr(h)]
        run = {hiplib.OP_CONV: self.yh_conv2d_fwd, hiplib.OP_STEM: self.yh_conv2d_stem_fwd,
               hiplib.OP_POOL: self.yh_maxpool2d_fwd, hiplib.OP_COPY: self.yh_copy_channels,
               hiplib.OP_ADD: self.yh_add_channels, hiplib.OP_DECODE: self.yh_yolo_decode,
               hiplib.OP_DW: self.yh_dwconv2d_fwd, hiplib.OP_SE: self.yh_se_fwd, hiplib.OP_QCOPY: self.yh_qcopy,
               hiplib.OP_QPOOL: self.yh_qpool, hiplib.OP_QADD: self.yh_qadd,
               hiplib.OP_BN_STATS: self.yh_bn_stats, hiplib.OP_BN_FINALIZE: self.yh_bn_finalize,
               hiplib.OP_BN_ACT_FWD: self.yh_bn_act_fwd, hiplib.OP_BN_BWD_REDUCE: self.yh_bn_act_bwd_reduce,
               hiplib.OP_BN_BWD_APPLY: self.yh_bn_act_bwd_apply, hiplib.OP_WGRAD: self.yh_conv2d_wgrad,
               hiplib.OP_STEM_WGRAD: self.yh_stem_wgrad, hiplib.OP_DILATE2: self.yh_dilate2,
               hiplib.OP_UPSAMPLE2_BWD: self.yh_upsample2_bwd, hiplib.OP_CAST_F32: self.yh_cast_f32,
               hiplib.OP_NCHW_TO_NHWC: self._layout, hiplib.OP_POOL_BWD: self.yh_maxpool2d_bwd,
               hiplib.OP_PACK_BATCH: lambda d, st: self.yh_pack_batch(d.items, d.n_items, st),
               hiplib.OP_STEM_BWD: self.yh_stem_bwd,
               hiplib.OP_DW_WGRAD: self.yh_dw_wgrad, hiplib.OP_DW_DGRAD: self.yh_dw_dgrad, hiplib.OP_SE_BWD: self.yh_se_bwd}
        lanes, deps = plan.get('lane', {}), plan.get('deps', {})
        pending, done_main = [], set()

        def execute(i):
            kind, desc, fixups = plan['ops'][i]
            d = type(desc).from_buffer_copy(bytes(desc))
            for off, slot, boff in fixups:
                base = plan['slots'][slot]
                assert base, 'slot %d was never bound' % slot
                C.c_void_p.from_address(C.addressof(d) + off).value = None if base == hiplib.SLOT_NULL else base + boff
            rc = run[kind](d, stream)
            self.calls.append(kind)
            return rc

        def flush(upto):
            """Run the queued side-lane ops up to and including op `upto` (the side lane is a stream: in order)."""
            while pending and pending[0] <= upto:
                i = pending.pop(0)
                for dep in deps.get(i, []):
                    assert dep < first or lanes.get(dep, 0) == 1 or dep in done_main, 'side op %d runs before its dependency %d' % (i, dep)
                rc = execute(i)
                if rc:
                    return rc
            return 0
        for i in range(first, last):
            if lanes.get(i, 0) == 1:
                pending.append(i)
                continue
            for dep in deps.get(i, []):
                if dep >= first and lanes.get(dep, 0) == 1:
                    rc = flush(dep)
                    if rc:
                        return rc
            rc = execute(i)
            if rc:
                return rc
            done_main.add(i)
        return flush(last)
