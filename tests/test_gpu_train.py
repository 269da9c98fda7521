"""GPU tier, training path: the backward / train-mode kernels against the host emulation AND against
torch autograd, then whole training steps through ``models.Darknet`` on the GPU against eager fp32 autograd on CPU.

Tolerances: fp32 mode (exact-product MFMA 16x16x4, fp32 statistics) agrees with fp32 autograd to round-off (1e-5
relative); fp16 storage is one rounding per stored tensor with fp32 accumulation, bounded per kernel below.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import conftest
import fakelib
import ops_harness as oh
import synth
import train_harness as th
from engine import hiplib
from engine.hiplib import ResampleDesc, CastDesc, PoolBwdDesc

pytestmark = pytest.mark.gpu

F16, F32 = hiplib.YH_F16, hiplib.YH_F32
DRY = bool(os.environ.get('YOLO_TEST_DRY_GPU'))
GPU = 'cpu' if DRY else 'cuda'
P = hiplib.ptr


@pytest.fixture(scope='module')
def libs():
    if DRY:
        return fakelib.FakeLib(), fakelib.FakeLib()
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return hiplib.load(), fakelib.FakeLib()


def _sync():
    if GPU == 'cuda':
        torch.cuda.synchronize()


@pytest.mark.parametrize('two_stage', [False, True], ids=['atomics', 'workspace'])
@pytest.mark.parametrize('code', [F32, F16], ids=['fp32', 'fp16'])
@pytest.mark.parametrize('case', [(3, 17, 19, 64, 1, 1, False, True), (2, 8, 8, 256, 5, 2, False, True),
                                  (2, 12, 12, 32, 1, 1, True, True), (2, 9, 9, 24, 4, 1, False, False)],
                         ids=['leaky', 'mish_ups', 'leaky_res', 'hswish_nobn'])
def test_bn_train_kernels(libs, code, case, two_stage):
    """stats -> finalize -> act forward -> backward reduce -> backward apply, vs emulation and vs torch autograd."""
    lib, fake = libs
    N, H, W, c, act, ups, use_res, use_bn = case
    g = torch.Generator().manual_seed(c * 7 + act)
    dt = oh.tdtype(code)
    z0 = (torch.randn(N, H, W, c + 8, generator=g) * 1.5 + 0.3).to(dt)       # channels [8, 8+c) of a wider buffer
    res0 = torch.randn(N, H, W, c, generator=g).to(dt) if use_res else None
    dy0 = torch.randn(N, H * ups, W * ups, c + 16, generator=g).to(dt)
    gamma0, beta0 = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    outs = []
    for L, dev in ((lib, GPU), (fake, 'cpu')):
        mv = lambda t: None if t is None else t.to(dev).clone()
        z, res, gamma, beta = mv(z0), mv(res0), mv(gamma0) if use_bn else None, mv(beta0)
        f32 = lambda n=c: torch.zeros(n, device=dev, dtype=torch.float32)
        mean, invstd, s1, s2, rm, rv = f32(), f32(), f32(), f32(), f32(), f32() + 1
        y = torch.full((N, H * ups, W * ups, c + 16), 3.0, device=dev, dtype=dt)
        kw = dict(gamma=gamma, beta=beta, mean=mean, invstd=invstd, act=act)
        WS = (lambda d: oh.with_bn_workspace(L, d)) if (two_stage and L is lib and not DRY) else (lambda d: d)
        if use_bn:
            oh.call(L, 'yh_bn_stats', WS(oh.bn_desc(code, z, c, 8, s1=s1, s2=s2, **kw)))
            oh.call(L, 'yh_bn_finalize', oh.bn_desc(code, z, c, 8, s1=s1, s2=s2, rmean=rm, rvar=rv, **kw))
        oh.call(L, 'yh_bn_act_fwd', oh.bn_desc(code, z, c, 8, res=res, out=y, out_off=16, ups=ups, **kw))
        # backward at the un-upsampled resolution (the engine reduces the fused upsample first)
        dy = mv(dy0)[:, ::ups, ::ups].contiguous()
        dbeta, dgamma = f32(), f32()
        dz = torch.full((N, H, W, c), 3.0, device=dev, dtype=dt)
        dyv = dy[..., 16:].contiguous()
        oh.call(L, 'yh_bn_act_bwd_reduce', WS(oh.bn_desc(code, z, c, 8, dy=dyv, s1=dbeta, s2=dgamma, **kw)))
        oh.call(L, 'yh_bn_act_bwd_apply', oh.bn_desc(code, z, c, 8, dy=dyv, out=dz, s1=dbeta, s2=dgamma, **kw))
        _sync()
        outs.append([t.float().cpu() for t in (mean, invstd, rm, rv, y, dbeta, dgamma, dz)])
    tol = 1e-5 if code == F32 else 2e-3
    names = ['mean', 'invstd', 'running_mean', 'running_var', 'y', 'dbeta', 'dgamma', 'dz']
    for name, a, b in zip(names, *outs):
        assert (a - b).abs().max().item() <= tol * (b.abs().max().item() + 1e-3), name
    assert torch.equal(outs[0][4][..., :16], torch.full_like(outs[0][4][..., :16], 3.0)), 'wrote outside the slice'
    # independent check against torch autograd on the same (rounded) inputs
    zz = z0[..., 8:8 + c].float().permute(0, 3, 1, 2).requires_grad_()
    ga, be = gamma0.clone().requires_grad_(), beta0.clone().requires_grad_()
    if use_bn:
        u = F.batch_norm(zz, None, None, ga, be, True, 0.1, 1e-5)
    else:
        u = zz + be.view(1, -1, 1, 1)
    yy = fakelib._act(u, act, 0.1)
    dyr = dy0[:, ::ups, ::ups, 16:].float().permute(0, 3, 1, 2)
    yy.backward(dyr)
    dz_ref = zz.grad.permute(0, 2, 3, 1)
    rt = 2e-4 if code == F32 else 4e-3
    assert (outs[0][7] - dz_ref).abs().max().item() <= rt * dz_ref.abs().max().item()
    assert (outs[0][5] - be.grad).abs().max().item() <= rt * be.grad.abs().max().item()
    if use_bn:
        assert (outs[0][6] - ga.grad).abs().max().item() <= rt * ga.grad.abs().max().item()


WGRAD_CASES = [
    # N, H, W, cin, cout, k, s, x_extra, dz_extra
    (2, 16, 16, 64, 128, 1, 1, 0, 0),
    (2, 20, 20, 32, 64, 3, 1, 0, 0),
    (1, 33, 31, 64, 128, 3, 2, 0, 0),
    (3, 19, 19, 128, 255, 1, 1, 0, 0),      # head: cout 255 in a 256-channel dz
    (1, 10, 10, 512, 1024, 3, 1, 0, 0),     # many tiles
    (2, 12, 12, 96, 72, 3, 1, 32, 16),      # pitched slices, partial tiles both ways
    (4, 64, 64, 16, 32, 3, 1, 0, 0),        # long pixel axis -> many splits
    (2, 26, 26, 256, 512, 3, 2, 0, 0),
    (2, 18, 18, 24, 40, 3, 1, 0, 0),        # 64-row tile, 216 flattened (tap, ci) columns
    (2, 16, 16, 64, 32, 1, 1, 0, 0),
    (1, 38, 38, 32, 64, 3, 2, 8, 8),
    # 3x3 / s1 halo form (fp16 with a workspace, cout % 256 == 0, cin % 32 == 0, 16 <= W <= 126; otherwise the im2col kernels)
    (2, 20, 20, 64, 256, 3, 1, 0, 0),
    (3, 19, 19, 32, 256, 3, 1, 0, 0),
    (1, 38, 38, 128, 512, 3, 1, 0, 0),
    (2, 17, 23, 96, 256, 3, 1, 32, 16),     # non-square, pitched slices, three ci tiles
    (5, 16, 16, 32, 256, 3, 1, 0, 0),       # several images (and their shared pad rows) per 512-pixel chunk
]


@pytest.mark.parametrize('use_ws', [True, False], ids=['workspace', 'atomics'])
@pytest.mark.parametrize('code', [F32, F16], ids=['fp32', 'fp16'])
@pytest.mark.parametrize('case', WGRAD_CASES, ids=lambda c: 'n%d_%dx%d_c%d-%d_k%ds%d_x%d_z%d' % c)
def test_wgrad_matches_autograd(libs, code, case, use_ws):
    lib, fake = libs
    N, H, W, cin, cout, k, s, xe, ze = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    dt = oh.tdtype(code)
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    cout_phys = oh.round_up(cout, 8)
    x = torch.randn(N, H, W, cin + xe, generator=g).to(dt)
    dz = torch.randn(N, Ho, Wo, cout_phys + ze, generator=g).to(dt)
    dz[..., ze + cout:] = 0
    got = oh.wgrad(lib, code, x.to(GPU), dz.to(GPU), cin, cout, k, s, pad, x_off=xe, dz_off=ze, use_ws=use_ws)
    _sync()
    emu = oh.wgrad(fake, code, x.clone(), dz.clone(), cin, cout, k, s, pad, x_off=xe, dz_off=ze)
    xr = x[..., xe:].float().permute(0, 3, 1, 2)
    dzr = dz[..., ze:ze + cout].float().permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(xr, (cout, cin, k, k), dzr, stride=s, padding=pad)
    scale = ref.abs().max().item()
    assert (emu - ref).abs().max().item() <= 1e-4 * scale
    # operands are identical on both sides; only the fp32 accumulation order differs
    assert (got.cpu() - ref).abs().max().item() <= (2e-5 if code == F32 else 1e-4) * scale


@pytest.mark.parametrize('use_ws', [True, False], ids=['workspace', 'atomics'])
@pytest.mark.parametrize('bm', ['128', '256'])
@pytest.mark.parametrize('case', [(2, 21, 19, 40, 256, 3, 1), (3, 16, 16, 64, 512, 1, 1), (1, 33, 31, 24, 256, 3, 2)],
                         ids=lambda c: 'n%d_%dx%d_c%d-%d_k%ds%d' % c)
def test_wgrad_row_tiles_agree(libs, monkeypatch, case, bm, use_ws):
    """Both row tiles of the LDS-DMA kernel (128 and 256 output channels per workgroup; the library picks by layer size,
    YH_WGRAD_BM forces one) against autograd, exact on small-integer operands, with ragged pixel counts and several splits."""
    if DRY:
        pytest.skip('kernel-only property')
    lib, _ = libs
    monkeypatch.setenv('YH_WGRAD_BM', bm)
    N, H, W, cin, cout, k, s = case
    g = torch.Generator().manual_seed(sum(case))
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randint(-3, 4, (N, H, W, cin), generator=g).half()
    dz = torch.randint(-2, 3, (N, Ho, Wo, cout), generator=g).half()
    got = oh.wgrad(lib, F16, x.to(GPU), dz.to(GPU), cin, cout, k, s, pad, use_ws=use_ws)
    _sync()
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (cout, cin, k, k), dz.float().permute(0, 3, 1, 2),
                                      stride=s, padding=pad)
    assert torch.equal(got.cpu(), ref)


@pytest.mark.parametrize('mode', ['1', '2', '3'], ids=['halo', 'auto', 'roll'])
@pytest.mark.parametrize('wgs', ['', '4', '64'], ids=['auto', 'few_splits', 'many_splits'])
@pytest.mark.parametrize('case', [(2, 21, 19, 64, 256), (7, 16, 40, 32, 512), (1, 76, 76, 128, 256), (3, 33, 152, 32, 256),
                                  (2, 23, 17, 128, 128), (1, 9, 152, 64, 128), (1, 5, 190, 64, 384), (9, 4, 16, 192, 128)],
                         ids=lambda c: 'n%d_%dx%d_c%d-%d' % c)
def test_wgrad_halo_form_is_exact_on_small_integers(libs, monkeypatch, case, wgs, mode):
    """The 3x3 halo weight-gradient kernels on integer operands: every product and partial sum exact, so any mistake in the pad
    handling, the tap offsets, the split ranges or the partial-tile reduction shows as a wrong integer.  mode 1: conv_wgrad_halo_kernel
    (csrc/conv_wgrad.hip: one shared pad row / column, dz in virtual pixel order, halo image read as nine shifted views; cout % 256,
    cin % 32); mode 3: conv_wgrad_roll_kernel wherever it qualifies (csrc/conv_wgrad_roll.hip: 128 x [9 x 64] tile, x rows rolling
    through a 512-row ring; cout % 128, cin % 64); mode 2 (the default): the library's per-layer choice between the two - a case one
    form does not take runs on the next form down, which must be just as exact.  Ragged last
    chunks / steps, split counts that do not divide, W = 152 and 190 (the shortest dz rings), images smaller than one 32-position step."""
    if DRY:
        pytest.skip('kernel-only property')
    lib, _ = libs
    monkeypatch.setenv('YH_WGRAD_HALO', mode)
    if wgs:
        monkeypatch.setenv('YH_WGRAD_HALO_WGS', wgs)
    N, H, W, cin, cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randint(-3, 4, (N, H, W, cin), generator=g).half()
    dz = torch.randint(-2, 3, (N, H, W, cout), generator=g).half()
    got = oh.wgrad(lib, F16, x.to(GPU), dz.to(GPU), cin, cout, 3, 1, 1, use_ws=True)
    _sync()
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (cout, cin, 3, 3), dz.float().permute(0, 3, 1, 2), padding=1)
    assert torch.equal(got.cpu(), ref)


@pytest.mark.parametrize('mode,kernel', [('1', 90), ('3', 91)], ids=['halo', 'roll'])
@pytest.mark.parametrize('case', [(2, 21, 19, 64, 256), (1, 76, 76, 128, 256), (3, 38, 38, 256, 512), (3, 33, 120, 64, 256)],
                         ids=lambda c: 'n%d_%dx%d_c%d-%d' % c)
def test_wgrad_halo_form_matches_autograd_on_random_operands(libs, monkeypatch, case, mode, kernel):
    """the halo kernels against torch autograd DIRECTLY (float64 conv2d weight gradient of the same f16
    operands), random - not small-integer - values: products exact in fp32, accumulation order the only difference (<= 1e-4 of the
    gradient's scale, the bound of the im2col form in test_wgrad_matches_autograd)."""
    if DRY:
        pytest.skip('kernel-only property')
    lib, _ = libs
    monkeypatch.setenv('YH_WGRAD_HALO', mode)
    N, H, W, cin, cout = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = (torch.randn(N, H, W, cin, generator=g) * 0.7).half()
    dz = (torch.randn(N, H, W, cout, generator=g) * 0.05).half()
    d = oh.WgradDesc(n=N, h=H, w_in=W, cin=cin, ho=H, wo=W, cout=cout, kh=3, kw=3, stride=1, pad=1, ldx=cin, lddz=cout, dtype=F16, splits=0)
    d.x = d.dz = d.dw = 4096
    assert lib.yh_conv2d_wgrad_kernel(oh.C.byref(d)) == kernel, 'the case must run on the kernel under test'
    got = oh.wgrad(lib, F16, x.to(GPU), dz.to(GPU), cin, cout, 3, 1, 1, use_ws=True).cpu().double()
    ref = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2), (cout, cin, 3, 3), dz.double().permute(0, 3, 1, 2), padding=1)
    err = (got - ref).abs().max().item()
    assert err <= 1e-4 * ref.abs().max().item(), (err, ref.abs().max().item())


@pytest.mark.parametrize('case', [(2, 17, 23, 64, 128, 32, 16), (1, 20, 31, 128, 256, 64, 128)], ids=lambda c: 'n%d_%dx%d_c%d-%d_x%d_z%d' % c)
def test_wgrad_roll_form_reads_pitched_channel_slices(libs, monkeypatch, case):
    """conv_wgrad_roll_kernel on operands that are channel slices of wider buffers (route / concat outputs: ldx > cin, lddz > cout,
    non-zero first channel), exact on integers; the surrounding channels hold large values that must not be read."""
    if DRY:
        pytest.skip('kernel-only property')
    lib, _ = libs
    monkeypatch.setenv('YH_WGRAD_HALO', '3')
    N, H, W, cin, cout, xe, ze = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randint(-3, 4, (N, H, W, cin + 2 * xe), generator=g).half()
    dz = torch.randint(-2, 3, (N, H, W, cout + 2 * ze), generator=g).half()
    x[..., :xe] = 1000.0
    x[..., xe + cin:] = 1000.0
    dz[..., :ze] = 1000.0
    dz[..., ze + cout:] = 1000.0
    got = oh.wgrad(lib, F16, x.to(GPU), dz.to(GPU), cin, cout, 3, 1, 1, x_off=xe, dz_off=ze, use_ws=True)
    _sync()
    ref = torch.nn.grad.conv2d_weight(x[..., xe:xe + cin].float().permute(0, 3, 1, 2), (cout, cin, 3, 3),
                                      dz[..., ze:ze + cout].float().permute(0, 3, 1, 2), padding=1)
    assert torch.equal(got.cpu(), ref)


def test_wgrad_f16_exact_on_small_integers(libs):
    """Integer operands: every product and partial sum is exact, so MFMA + atomics must reproduce autograd bit for bit."""
    lib, _ = libs
    g = torch.Generator().manual_seed(3)
    x = torch.randint(-3, 4, (2, 14, 14, 64), generator=g).half()
    dz = torch.randint(-2, 3, (2, 14, 14, 128), generator=g).half()
    got = oh.wgrad(lib, F16, x.to(GPU), dz.to(GPU), 64, 128, 3, 1, 1)
    _sync()
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (128, 64, 3, 3), dz.float().permute(0, 3, 1, 2), padding=1)
    assert torch.equal(got.cpu(), ref)


@pytest.mark.parametrize('code', [F32, F16], ids=['fp32', 'fp16'])
@pytest.mark.parametrize('shape', [(2, 32, 32, 32, 1), (3, 21, 19, 16, 1), (1, 40, 40, 32, 2)], ids=str)
def test_stem_wgrad_matches_autograd(libs, code, shape):
    lib, _ = libs
    N, H, W, cout, s = shape
    g = torch.Generator().manual_seed(H)
    dt = oh.tdtype(code)
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    x = torch.rand(N, 3, H, W, generator=g)
    dz = torch.randn(N, Ho, Wo, cout, generator=g).to(dt)
    got = oh.stem_wgrad(lib, code, x.to(GPU), dz.to(GPU), cout, stride=s)
    _sync()
    ref = torch.nn.grad.conv2d_weight(x, (cout, 3, 3, 3), dz.float().permute(0, 3, 1, 2), stride=s, padding=1)
    assert (got.cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    # the route the engine takes: 8-channel NHWC copy of the image + the MFMA kernel with cin_w = 3
    got2, img = oh.stem_wgrad_mfma(lib, code, x.to(GPU), dz.to(GPU), cout, stride=s)
    _sync()
    assert torch.equal(img[..., :3].float().cpu(), x.permute(0, 2, 3, 1).to(dt).float()) and img[..., 3:].abs().max() == 0
    ref2 = torch.nn.grad.conv2d_weight(x.to(dt).float(), (cout, 3, 3, 3), dz.float().permute(0, 3, 1, 2), stride=s, padding=1)
    assert (got2.cpu() - ref2).abs().max().item() <= 1e-4 * ref2.abs().max().item()


DGRAD_CASES = [(2, 16, 16, 64, 128, 1, 1), (2, 20, 20, 32, 64, 3, 1), (2, 32, 32, 64, 128, 3, 2), (1, 25, 27, 32, 64, 3, 2),
               (2, 13, 13, 128, 255, 1, 1), (1, 10, 10, 256, 512, 3, 1)]


@pytest.mark.parametrize('case', [(2, 3, 40, 64, 32, 1), (1, 3, 21, 45, 32, 1), (3, 1, 17, 33, 16, 5), (2, 3, 9, 100, 16, 1), (1, 3, 96, 608, 32, 1)],
                         ids=['c32', 'c32_tail', 'gray_c16_mish', 'c16_wide', 'row608'])
def test_first_block_backward_in_one_pass(libs, case):
    """csrc/stem_bwd.hip: dgamma, dbeta and dW of block 0 from ONE pass over dy and z (closed form over Q = g X^T, R = xhat X^T,
    SX), against (a) float64 torch: train-mode batch_norm backward + conv2d weight gradient through autograd on the SAME f16 dy / z
    (reference: autograd through models.py:92-113), and (b) the emulation of the descriptor.  fp16 MFMA operands, fp32 accumulation:
    agreement to the operand rounding (2^-11 relative per term, averaged over thousands of pixels)."""
    lib, fake = libs
    N, cin, H, W, c, act = case
    g = torch.Generator().manual_seed(H * 13 + W)
    x = torch.rand(N, cin, H, W, generator=g)
    z = (torch.randn(N, H, W, c + 8, generator=g) * 1.3 + 0.2).half()        # channels [0, c) of wider buffers: pitches > c
    dy = (torch.randn(N, H, W, c + 16, generator=g) * 0.05).half()
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    zz = z[..., :c].float().reshape(-1, c).double()
    mean = zz.mean(0)
    invstd = 1.0 / torch.sqrt(zz.var(0, unbiased=False) + 1e-5)
    # float64 reference through autograd: y = act(bn(conv)); the conv output is GIVEN (z), so differentiate bn + act w.r.t. z and
    # push that gradient into conv2d's weight gradient
    zt = zz.clone().requires_grad_(True)
    xh = (zt - zt.mean(0)) / torch.sqrt(zt.var(0, unbiased=False) + 1e-5)
    ga, be = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    u = ga * xh + be
    y = {1: lambda t: F.leaky_relu(t, 0.1), 5: lambda t: t * torch.tanh(F.softplus(t))}[act](u)
    dyd = dy[..., :c].float().reshape(-1, c).double()
    dz, dga, dbe = torch.autograd.grad((y * dyd).sum(), (zt, ga, be))
    dzn = dz.view(N, H, W, c).permute(0, 3, 1, 2)
    wref = torch.nn.grad.conv2d_weight(x.double(), (c, cin, 3, 3), dzn, stride=1, padding=1)
    outs = []
    for L, dev in ((lib, GPU), (fake, 'cpu')):
        t = lambda a: a.to(dev)
        outs.append([o.cpu() for o in oh.stem_bwd(L, t(x), t(dy), t(z), t(gamma), t(beta), t(mean.float()), t(invstd.float()), act=act)])
    (dw, dg, db), (dw_e, dg_e, db_e) = outs
    for got, emu, want, tag in ((dw, dw_e, wref, 'dw'), (dg, dg_e, dga, 'dgamma'), (db, db_e, dbe, 'dbeta')):
        scale = want.abs().max().item()
        assert (got.double() - want).abs().max().item() <= 3e-3 * scale, (tag, (got.double() - want).abs().max().item(), scale)
        assert (got - emu).abs().max().item() <= 1e-3 * scale, (tag, 'vs emulation')


@pytest.mark.parametrize('case', [(2, 3, 40, 64, 1), (1, 3, 21, 45, 1), (3, 1, 17, 33, 5), (1, 3, 96, 608, 1), (2, 3, 8, 31, 1)],
                         ids=['even', 'odd_sizes_tail', 'gray_mish', 'row608', 'odd_w'])
def test_first_block_backward_with_the_next_data_gradient_fused_in(libs, case):
    """yh_stem_bwd with dz1 / w1: dy of block 0 is never stored - the pass computes conv_transpose(dz1, W1) (the data gradient of a
    3x3 / s2 / p1 conv with 64 output channels) per segment.  Against (a) float64 torch: conv_transpose2d -> rounded to f16 like the
    stored tensor -> BatchNorm / activation backward through autograd -> conv2d weight gradient, (b) the emulation."""
    lib, fake = libs
    N, cin, H, W, act = case
    c, k1 = 32, 64
    H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.rand(N, cin, H, W, generator=g)
    z = (torch.randn(N, H, W, c, generator=g) * 1.3 + 0.2).half()
    dz1 = (torch.randn(N, H1, W1, k1 + 8, generator=g) * 0.05).half()
    w1 = torch.randn(k1, c, 3, 3, generator=g) * (c * 9) ** -0.5
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    zz = z.float().reshape(-1, c).double()
    mean, invstd = zz.mean(0), 1.0 / torch.sqrt(zz.var(0, unbiased=False) + 1e-5)
    op = (H - (2 * (H1 - 1) + 1), W - (2 * (W1 - 1) + 1))
    dy = F.conv_transpose2d(dz1[..., :k1].float().permute(0, 3, 1, 2).double(), w1.half().double(), stride=2, padding=1, output_padding=op)
    dyd = dy.permute(0, 2, 3, 1).reshape(-1, c).half().double()          # what the separate launch would have stored
    zt = zz.clone().requires_grad_(True)
    xh = (zt - zt.mean(0)) / torch.sqrt(zt.var(0, unbiased=False) + 1e-5)
    ga, be = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    u = ga * xh + be
    y = {1: lambda t: F.leaky_relu(t, 0.1), 5: lambda t: t * torch.tanh(F.softplus(t))}[act](u)
    dz, dga, dbe = torch.autograd.grad((y * dyd).sum(), (zt, ga, be))
    wref = torch.nn.grad.conv2d_weight(x.double(), (c, cin, 3, 3), dz.view(N, H, W, c).permute(0, 3, 1, 2), stride=1, padding=1)
    outs = []
    for L, dev in ((lib, GPU), (fake, 'cpu')):
        t = lambda a: a.to(dev)
        outs.append([o.cpu() for o in oh.stem_bwd(L, t(x), None, t(z), t(gamma), t(beta), t(mean.float()), t(invstd.float()), act=act,
                                                  dz1=t(dz1), w1=t(w1))])
    (dw, dg, db), (dw_e, dg_e, db_e) = outs
    for got, emu, want, tag in ((dw, dw_e, wref, 'dw'), (dg, dg_e, dga, 'dgamma'), (db, db_e, dbe, 'dbeta')):
        scale = want.abs().max().item()
        assert (got.double() - want).abs().max().item() <= 4e-3 * scale, (tag, (got.double() - want).abs().max().item(), scale)
        assert (got - emu).abs().max().item() <= 2e-3 * scale, (tag, 'vs emulation')


@pytest.mark.parametrize('code', [F32, F16], ids=['fp32', 'fp16'])
@pytest.mark.parametrize('accumulate', [False, True], ids=['write', 'acc'])
@pytest.mark.parametrize('case', DGRAD_CASES, ids=lambda c: 'n%d_%dx%d_c%d-%d_k%ds%d' % c)
def test_dgrad_matches_autograd(libs, code, accumulate, case):
    """conv(dz [dilated], transposed+flipped weights) == autograd's input gradient, incl. odd sizes under stride 2."""
    lib, _ = libs
    N, H, W, cin, cout, k, s = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    dt = oh.tdtype(code)
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    cphys = oh.round_up(cout, 8)
    w = torch.randn(cout, cin, k, k, generator=g) * (cout * k * k) ** -0.5
    dz = torch.randn(N, Ho, Wo, cphys, generator=g).to(dt)
    dz[..., cout:] = 0
    acc0 = torch.randn(N, H, W, cin, generator=g).to(dt) if accumulate else None
    acc = None if acc0 is None else acc0.to(GPU).clone()
    dx = oh.dgrad(lib, code, dz.to(GPU), w.to(GPU), (H, W), s, pad, acc=acc)
    _sync()
    wq = w.to(dt).float()
    ref = torch.nn.grad.conv2d_input((N, cin, H, W), wq, dz[..., :cout].float().permute(0, 3, 1, 2), stride=s, padding=pad)
    ref = ref.permute(0, 2, 3, 1)
    if accumulate:
        ref = ref + acc0.float()
    tol = 2e-5 if code == F32 else 2.5e-3
    assert (dx.float().cpu() - ref).abs().max().item() <= tol * ref.abs().max().item()


@pytest.mark.parametrize('code', [F32, F16], ids=['fp32', 'fp16'])
@pytest.mark.parametrize('accumulate', [False, True], ids=['write', 'acc'])
@pytest.mark.parametrize('case', [(2, 32, 32, 32, 64), (1, 25, 27, 32, 64), (3, 19, 22, 16, 40), (1, 9, 9, 24, 64), (2, 21, 24, 64, 128)],
                         ids=lambda c: 'n%d_%dx%d_c%d-%d' % c)
def test_fused_stride2_dgrad_matches_autograd_and_the_phase_form(libs, code, accumulate, case):
    """ups = 4: the four parity phases of a 3x3 / stride-2 data gradient as one 2x2-tap pass with 4 * cin rows (the form the
    engine uses for layers with <= 64 input channels), against autograd and against the four-launch phase form, odd sizes
    included; also through the host emulation."""
    lib, fake = libs
    N, H, W, cin, cout = case
    g = torch.Generator().manual_seed(sum(case))
    dt = oh.tdtype(code)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    cphys = oh.round_up(cout, 8)
    w = torch.randn(cout, cin, 3, 3, generator=g) * (cout * 9) ** -0.5
    dz = torch.randn(N, Ho, Wo, cphys, generator=g).to(dt)
    dz[..., cout:] = 0
    acc0 = torch.randn(N, H, W, cin, generator=g).to(dt) if accumulate else None
    outs = []
    for L, dev, fused in ((lib, GPU, True), (lib, GPU, False), (fake, 'cpu', True)):
        acc = None if acc0 is None else acc0.to(dev).clone()
        outs.append(oh.dgrad(L, code, dz.to(dev), w.to(dev), (H, W), 2, 1, acc=acc, fused=fused).float().cpu())
        _sync()
    ref = torch.nn.grad.conv2d_input((N, cin, H, W), w.to(dt).float(), dz[..., :cout].float().permute(0, 3, 1, 2), stride=2, padding=1)
    ref = ref.permute(0, 2, 3, 1)
    if accumulate:
        ref = ref + acc0.float()
    tol = 2e-5 if code == F32 else 2.5e-3
    for got in outs:
        assert (got - ref).abs().max().item() <= tol * ref.abs().max().item()


@pytest.mark.parametrize('code', [F32, F16], ids=['fp32', 'fp16'])
def test_resample_and_cast_kernels(libs, code):
    lib, fake = libs
    g = torch.Generator().manual_seed(9)
    dt = oh.tdtype(code)
    N, H, W, c = 2, 7, 9, 24
    big = torch.randn(N, 2 * H, 2 * W, c + 8, generator=g).to(dt)
    src32 = torch.randn(N, H, W, c + 4, generator=g)
    outs = []
    for L, dev in ((lib, GPU), (fake, 'cpu')):
        b = big.to(dev)
        small = torch.full((N, H, W, c), 3.0, device=dev, dtype=dt)
        oh.call(L, 'yh_upsample2_bwd', ResampleDesc(x=P(b, 8), y=P(small), n=N, h=H, w_in=W, c=c, big_h=2 * H, big_w=2 * W,
                                                    ldx=c + 8, ldy=c, dtype=code))
        dil = torch.zeros((N, 2 * H - 1, 2 * W, c), device=dev, dtype=dt)      # odd height, even width
        oh.call(L, 'yh_dilate2', ResampleDesc(x=P(small), y=P(dil), n=N, h=H, w_in=W, c=c, big_h=2 * H - 1, big_w=2 * W,
                                              ldx=c, ldy=c, dtype=code))
        s32 = src32.to(dev)
        cast = torch.full((N, H, W, c), 3.0, device=dev, dtype=dt)
        oh.call(L, 'yh_cast_f32', CastDesc(x=P(s32), y=P(cast), pixels=N * H * W, c=c, ldx=c + 4, ldy=c, dtype=code))
        _sync()
        outs.append([t.float().cpu() for t in (small, dil, cast)])
    for a, b in zip(*outs):
        assert (a - b).abs().max().item() <= (1e-6 if code == F32 else 2e-3) * b.abs().max().item()
    small_ref = big[..., 8:].float().view(N, H, 2, W, 2, c).sum((2, 4))
    assert (outs[0][0] - small_ref).abs().max().item() <= (1e-5 if code == F32 else 4e-3) * small_ref.abs().max().item()
    assert torch.equal(outs[0][1][:, ::2, ::2], outs[0][0]) and outs[0][1][:, 1::2].abs().max() == 0 \
        and outs[0][1][:, :, 1::2].abs().max() == 0
    assert torch.equal(outs[0][2], src32[..., :c].to(dt).float())


@pytest.mark.parametrize('code', [F32, F16], ids=['fp32', 'fp16'])
@pytest.mark.parametrize('case', [(2, 16, 16, 32, 2, 2, 0), (2, 13, 13, 64, 2, 1, 1), (2, 19, 19, 32, 5, 1, 0), (1, 19, 19, 16, 13, 1, 0),
                                  (2, 15, 17, 24, 9, 1, 0)], ids=lambda c: 'n%d_%dx%d_c%d_k%ds%d_z%d' % c)
def test_maxpool_backward_matches_autograd(libs, code, case):
    """Gradient goes to the first maximum of each window (torch's recorded index); overlapping windows add up."""
    lib, fake = libs
    N, H, W, c, k, s, edge_zero = case
    g = torch.Generator().manual_seed(k * 31 + H)
    dt = oh.tdtype(code)
    if edge_zero:
        Ho, Wo, pad_lo = H, W, 0
    else:
        pad_lo = (k - 1) // 2 if s == 1 else 0
        Ho, Wo = (H + 2 * pad_lo - k) // s + 1, (W + 2 * pad_lo - k) // s + 1
    x = torch.randn(N, H, W, c + 8, generator=g).to(dt)
    dy = torch.randn(N, Ho, Wo, c, generator=g).to(dt)
    acc0 = torch.randn(N, H, W, c + 8, generator=g).to(dt)       # an earlier contribution already in dx
    outs = []
    for L, dev in ((lib, GPU), (fake, 'cpu')):
        xd, dyd, dx = x.to(dev), dy.to(dev), acc0.to(dev).clone()
        d = PoolBwdDesc(x=P(xd, 8), dy=P(dyd), dx=P(dx, 8), n=N, h=H, w_in=W, c=c, ho=Ho, wo=Wo, k=k, stride=s, pad_lo=pad_lo,
                        edge_zero=edge_zero, ldx=c + 8, lddy=c, lddx=c + 8, dtype=code)
        oh.call(L, 'yh_maxpool2d_bwd', d)
        _sync()
        outs.append(dx.float().cpu())
    assert torch.equal(outs[0][..., :8], acc0[..., :8].float()), 'wrote outside the channel slice'
    scale = outs[1].abs().max().item()
    # fp16: overlapping windows are gathered in fp32 and rounded once (round 6; the packed-half atomics of rounds 1 - 5 rounded after every one
    # of up to k*k adds and needed 2e-2 at k 9 / 13); the non-overlapping k2 s2 case adds one value per element
    tol16 = 2e-3
    assert (outs[0] - outs[1]).abs().max().item() <= (1e-5 if code == F32 else tol16) * scale
    xr = x[..., 8:].float().permute(0, 3, 1, 2).requires_grad_()
    y = F.max_pool2d(F.pad(xr, (0, 1, 0, 1)), k, s) if edge_zero else F.max_pool2d(xr, k, s, pad_lo)
    y.backward(dy.float().permute(0, 3, 1, 2))
    ref = xr.grad.permute(0, 2, 3, 1) + acc0[..., 8:].float()
    assert (outs[0][..., 8:] - ref).abs().max().item() <= (1e-5 if code == F32 else tol16) * scale


def _rand(g, *shape, scale=1.0):
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize('act', [1, 5], ids=['leaky', 'mish'])
@pytest.mark.parametrize('case', [(5, 240, 232, 32, 64, True), (3, 304, 300, 64, 128, True), (5, 240, 232, 128, 256, True), (5, 232, 240, 128, 256, False)],
                         ids=['32-64_res', '64-128_res', '128-256_res', '128-256_write'])
def test_data_gradient_carries_the_completed_blocks_backward_sums(libs, case, act):
    """Round 6 (yh_conv_desc.bwd_z, conv_pw_lds.hip modes 3 / 4): a 1x1 data gradient - with or without its residual accumulate - that
    completes the gradient dy of a BatchNorm + activation block also leaves that block's backward sums.  Against the two launches it
    replaces: the stored tensor is bit-identical to the plain data gradient's, and the rows, added up by yh_bn_act_bwd_reduce(nparts),
    equal the reduction pass over that tensor and z (same per-element arithmetic, other summation order: 2e-5 of sum |terms|).
    Twice in a row: the same bits (fixed partition of the pixels over the waves, no atomics)."""
    if DRY:
        pytest.skip('the emulator carries the sums on any 1x1 data gradient; the plan-level test is tests/test_train_emulated.py')
    lib, _ = libs
    N, H, W, cin, cout, use_res = case
    g = torch.Generator().manual_seed(cin + cout + act)
    w = _rand(g, cout, cin, 1, 1, scale=cin ** -0.5).to(GPU)
    cb = torch.zeros(cout, device=GPU)
    x = _rand(g, N, H, W, cin).half().to(GPU)
    res = _rand(g, N, H, W, cout).half().to(GPU) if use_res else None
    z = _rand(g, N, H, W, cout).half().to(GPU)
    gamma, beta = (torch.rand(cout, generator=g) + 0.5).to(GPU), (torch.randn(cout, generator=g) * 0.3).to(GPU)
    mean, invstd = (torch.randn(cout, generator=g) * 0.2).to(GPU), (torch.rand(cout, generator=g) + 0.7).to(GPU)
    packed, bias, cin_k, m_pad = oh.pack_conv(lib, F16, w, cb, None, cin_phys=cin)
    plain = oh.conv(lib, F16, x, packed, bias, cin_k, m_pad, cout, 1, 1, 0, act=0, res=res).clone()
    runs = []
    for _ in range(2):
        bw = dict(z=z, gamma=gamma, beta=beta, mean=mean, invstd=invstd, act=act)
        y = oh.conv(lib, F16, x, packed, bias, cin_k, m_pad, cout, 1, 1, 0, act=0, res=res, bwd=bw)
        assert y is not None and bw['rows'] > 0, 'the library does not carry the sums on this shape'
        s1, s2 = torch.zeros(cout, device=GPU), torch.zeros(cout, device=GPU)
        d = oh.bn_desc(F16, z, cout, dy=y, gamma=gamma, beta=beta, mean=mean, invstd=invstd, s1=s1, s2=s2, act=act)
        d.nparts, d.ws, d.ws_floats = bw['rows'], oh.P(bw['ws']), bw['ws'].numel()
        oh.call(lib, 'yh_bn_act_bwd_reduce', d)
        torch.cuda.synchronize()
        runs.append((y.clone(), s1.clone(), s2.clone()))
    assert torch.equal(runs[0][0], plain), 'the stored data gradient changed'
    assert all(torch.equal(a, b) for a, b in zip(runs[0], runs[1])), 'not reproducible'
    r1, r2 = torch.zeros(cout, device=GPU), torch.zeros(cout, device=GPU)
    oh.call(lib, 'yh_bn_act_bwd_reduce', oh.bn_desc(F16, z, cout, dy=plain, gamma=gamma, beta=beta, mean=mean, invstd=invstd, s1=r1, s2=r2, act=act))
    torch.cuda.synchronize()
    yf, zf = plain.float().reshape(-1, cout), z.float().reshape(-1, cout)
    xh = (zf - mean) * invstd
    scale1 = yf.abs().sum(0) + 1e-6
    scale2 = (yf * xh).abs().sum(0) + 1e-6
    assert ((runs[0][1] - r1).abs() / scale1).max().item() <= 2e-5 and ((runs[0][2] - r2).abs() / scale2).max().item() <= 2e-5


@pytest.mark.parametrize('code', [F32, F16], ids=['fp32', 'fp16'])
@pytest.mark.parametrize('case', [(2, 20, 20, 32, 64, 3, 1, 0), (1, 33, 31, 64, 128, 3, 2, 0), (3, 19, 19, 128, 256, 1, 1, 0),
                                  (4, 32, 32, 32, 32, 3, 1, 3), (2, 26, 26, 128, 256, 3, 1, 26), (2, 40, 40, 64, 128, 1, 1, 27),
                                  (2, 24, 24, 64, 64, 3, 1, 24), (2, 16, 16, 64, 128, 1, 1, 25), (2, 16, 16, 64, 128, 3, 1, 1),
                                  # halo ping-pong kernel: partial rows per (512 VIRTUAL pixels, wave); padding positions must not count
                                  (2, 26, 26, 128, 256, 3, 1, 43), (5, 9, 7, 64, 128, 3, 1, 43), (1, 76, 76, 32, 192, 3, 1, 43),
                                  # streaming 3x3 kernel: one row per wave of the launch, the tail's repeated last pixel counts once
                                  (3, 37, 41, 32, 64, 3, 1, 72), (2, 45, 43, 32, 64, 3, 2, 72), (2, 33, 31, 32, 32, 3, 1, 72),
                                  (1, 3, 5, 32, 32, 3, 1, 72),
                                  # persistent LDS-weights 1x1 kernel: one row per pixel stream (a wave, or the wave pair of a 256-channel layer)
                                  (3, 37, 41, 256, 128, 1, 1, 73), (2, 45, 43, 64, 32, 1, 1, 73), (2, 33, 31, 128, 256, 1, 1, 73),
                                  (1, 3, 5, 128, 64, 1, 1, 73)],
                         ids=lambda c: 'n%d_%dx%d_c%d-%d_k%ds%d_t%d' % c)
def test_conv_epilogue_batch_statistics(libs, code, case):
    """Training forward: the conv epilogue's partial sums + yh_bn_finalize(nparts) == statistics of the stored output."""
    lib, _ = libs
    N, H, W, cin, cout, k, s, tile = case
    if code == F32 and tile in (43, 72, 73):
        pytest.skip('fp16 / int8 kernel')
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    dt = oh.tdtype(code)
    pad = (k - 1) // 2
    w = (torch.randn(cout, cin, k, k, generator=g) * (cin * k * k) ** -0.5).to(GPU)
    x = (torch.randn(N, H, W, cin, generator=g) + 0.3).to(dt).to(GPU)
    packed, bias, cin_k, m_pad = oh.pack_conv(lib, code, w, None, None, cin_phys=cin)
    st = {}
    y = oh.conv(lib, code, x, packed, bias, cin_k, m_pad, cout, k, s, pad, act=0, tile=tile, stats=st)
    f32 = lambda: torch.zeros(cout, device=GPU, dtype=torch.float32)
    mean, invstd, s1, s2 = f32(), f32(), f32(), f32()
    pix = y.shape[0] * y.shape[1] * y.shape[2]
    d = oh.bn_desc(code, y, cout, mean=mean, invstd=invstd, s1=s1, s2=s2)
    d.nparts, d.ws, d.ws_floats = st['rows'], P(st['ws']), st['ws'].numel()
    oh.call(lib, 'yh_bn_finalize', d)
    _sync()
    q = y.float().reshape(pix, cout)
    ref_mean = q.mean(0)
    ref_var = q.var(0, unbiased=False)
    assert not torch.isnan(st['ws']).any() or DRY, 'a partial row was not written'
    assert (mean - ref_mean).abs().max().item() <= 2e-5 * (ref_mean.abs().max().item() + 1)
    assert ((1 / invstd ** 2 - 1e-5) - ref_var).abs().max().item() <= 1e-4 * ref_var.abs().max().item()


@pytest.mark.parametrize('code', [F32, F16], ids=['fp32', 'fp16'])
def test_pack_batch_equals_single_layer_packers(libs, code):
    """One launch for every weight image of the step == the per-layer entry points, bit for bit."""
    import ctypes as C
    from engine.hiplib import PackItem
    lib, _ = libs
    g = torch.Generator().manual_seed(17)
    dt = oh.tdtype(code)
    kstep = 32 if code == F16 else 16
    w1 = torch.randn(40, 24, 3, 3, generator=g).to(GPU)          # stride-1 layer: forward + dgrad image
    w2 = torch.randn(64, 40, 3, 3, generator=g).to(GPU)          # stride-2 layer: forward + four phases
    b2 = torch.randn(64, generator=g).to(GPU)
    w0 = torch.randn(32, 3, 3, 3, generator=g).to(GPU)           # first layer
    keep, items, singles = [], [], []

    def buf(n, dtype=dt, fill=7.0):
        t = torch.full((n,), fill, device=GPU, dtype=dtype)
        keep.append(t)
        return t

    def both(n, dtype=dt):
        return buf(n, dtype), buf(n, dtype)
    for w, bias, stride in ((w1, None, 1), (w2, b2, 2)):
        cout, cin, k, _ = w.shape
        cin_k, m_pad = oh.round_up(oh.round_up(cin, 8), kstep), oh.round_up(oh.round_up(cout, 8), 128)
        cout_k, dm_pad = oh.round_up(oh.round_up(cout, 8), kstep), oh.round_up(oh.round_up(cin, 8), 128)
        pa, pb = both(m_pad * 9 * cin_k)
        ba, bb = both(m_pad, torch.float32)
        geo = dict(dtype=code, cout=cout, cin=cin, kh=3, kw=3, pad=1)
        items.append(PackItem(w=P(w), bias=P(bias), packed=P(pa), bias_out=P(ba), mode=0, k_pad=cin_k, m_pad=m_pad, **geo))
        assert lib.yh_conv_pack_weights(code, P(w), P(bias), None, None, None, None, 0.0, None, cout, cin, 3, 3, cin_k, m_pad, P(pb),
                                        P(bb), oh.stream()) == 0
        singles += [(pa, pb), (ba, bb)]
        if stride == 1:
            da, db = both(dm_pad * 9 * cout_k)
            items.append(PackItem(w=P(w), packed=P(da), mode=1, k_pad=cout_k, m_pad=dm_pad, **geo))
            assert lib.yh_conv_pack_weights_dgrad(code, P(w), cout, cin, 3, 3, cout_k, dm_pad, P(db), oh.stream()) == 0
            singles.append((da, db))
        else:
            for a in (0, 1):
                for b in (0, 1):
                    taps = ((a + 1) // 2 + 1) * ((b + 1) // 2 + 1)
                    da, db = both(dm_pad * taps * cout_k)
                    items.append(PackItem(w=P(w), packed=P(da), mode=2, k_pad=cout_k, m_pad=dm_pad, pa=a, pb=b, **geo))
                    assert lib.yh_conv_pack_weights_dgrad_phase(code, P(w), cout, cin, 3, 3, 1, a, b, cout_k, dm_pad, P(db), None, None,
                                                                oh.stream()) == 0
                    singles.append((da, db))
    sa, sb = both(27 * 32, torch.float32)
    sba, sbb = both(32, torch.float32)
    items.append(PackItem(w=P(w0), packed=P(sa), bias_out=P(sba), mode=3, dtype=code, cout=32, cin=3, kh=3, kw=3, pad=1, cout_pad=32))
    assert lib.yh_stem_pack_weights(P(w0), None, None, None, None, None, 0.0, 32, 3, 3, 3, 32, P(sb), P(sbb), oh.stream()) == 0
    singles += [(sa, sb), (sba, sbb)]
    arr = (PackItem * len(items))(*items)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(GPU)
    assert lib.yh_pack_batch(P(table), len(items), oh.stream()) == 0
    _sync()
    for k, (a, b) in enumerate(singles):
        assert torch.equal(a, b), 'image %d differs' % k


@pytest.mark.parametrize('rel,size,nc', [('yolov3/yolov3.cfg', 320, 80), ('yolov3tiny/yolov3-tiny-hand.cfg', 416, 1),
                                         ('yolov3tiny/yolov3-tiny-hand.cfg', 416, 91)], ids=['coco80', 'hand1', 'wide91'])
def test_fused_loss_kernels_match_torch_loss(libs, rel, size, nc, tmp_path):
    """csrc/loss.hip vs the torch restatement of compute_loss run on the CPU (itself pinned by the reference goldens on the CPU
    tier): loss items to 1e-5, gradient of every raw head element to 1e-4 of its scale, through strided NHWC head views.
    The CPU is the comparison point because cells matched by several targets take the LAST target's objectness there (what
    the kernels reproduce); torch's index_put on a GPU picks an arbitrary one."""
    if DRY:
        pytest.skip('the loss glue is covered on the emulator by tests/test_loss_golden.py')
    from models import Darknet
    from utils import utils as U
    torch.manual_seed(0)
    path = os.path.join(conftest.PKG, 'cfg', rel)
    if nc == 91:      # a head wider than one workgroup row (na * no = 3 * 96 = 288 > 256), any dataset with nc > 80
        text = open(path).read().replace('classes=1', 'classes=91').replace('filters=18', 'filters=288')
        path = str(tmp_path / 'tiny91.cfg')
        open(path, 'w').write(text)
    model = Darknet(path, (size, size))
    model.nc, model.gr = nc, 0.6
    model.hyp = {'giou': 3.54, 'cls': 37.4, 'cls_pw': 1.3, 'obj': 64.3, 'obj_pw': 0.8, 'iou_t': 0.20, 'fl_gamma': 0.0}
    raws, targets = synth.loss_inputs(model, size, batch=4, seed=33, labels_per_image=12)
    results = []
    for fused in (True, False):
        dev = GPU if fused else torch.device('cpu')
        bases, views = [], []
        for r in raws:
            bs, na, ny, nx, no = r.shape
            base = torch.zeros(bs, ny, nx, oh.round_up(na * no, 8), device=dev)
            base[..., :na * no] = r.to(dev).permute(0, 2, 3, 1, 4).reshape(bs, ny, nx, na * no)
            base.requires_grad_()
            bases.append(base)
            views.append(base[..., :na * no].view(bs, ny, nx, na, no).permute(0, 3, 1, 2, 4))
        loss, items = U.compute_loss(views, targets.to(dev), model, fused=fused)
        (loss * 7.0).backward()
        results.append((items.cpu(), [b.grad.cpu() for b in bases]))
    (it_f, g_f), (it_t, g_t) = results
    assert torch.allclose(it_f, it_t, rtol=1e-5, atol=1e-6), (it_f, it_t)
    for a, b in zip(g_f, g_t):
        assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item()


@pytest.mark.parametrize('code', [F32, F16], ids=['fp32', 'fp16'])
@pytest.mark.parametrize('case', [(2, 20, 20, 32, 3, 1), (2, 21, 19, 24, 5, 2), (1, 26, 26, 96, 3, 2), (3, 13, 13, 240, 5, 1)],
                         ids=lambda c: 'n%d_%dx%d_c%d_k%ds%d' % c)
def test_depthwise_backward_matches_autograd(libs, code, case):
    import ctypes as C
    from engine.hiplib import DwBwdDesc
    lib, _ = libs
    N, H, W, c, k, s = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    dt = oh.tdtype(code)
    pad = (k - 1) // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    x = torch.randn(N, H, W, c, generator=g).to(dt)
    dz = torch.randn(N, Ho, Wo, c, generator=g).to(dt)
    w = torch.randn(c, 1, k, k, generator=g) * 0.3
    acc0 = torch.randn(N, H, W, c, generator=g).to(dt)
    xd, dzd = x.to(GPU), dz.to(GPU)
    packed = torch.empty(k * k * c, device=GPU, dtype=dt)
    bias = torch.empty(c, device=GPU, dtype=torch.float32)
    assert lib.yh_dw_pack_weights(code, P(w.to(GPU)), None, None, None, None, None, 0.0, None, c, k, c, P(packed), P(bias), oh.stream()) == 0
    dw = torch.zeros(c, 1, k, k, device=GPU)
    geo = dict(n=N, h=H, w_in=W, c=c, ho=Ho, wo=Wo, k=k, stride=s, pad=pad, ldx=c, lddz=c, dtype=code)
    oh.call(lib, 'yh_dw_wgrad', DwBwdDesc(x=P(xd), dz=P(dzd), dw=P(dw), lddx=0, accumulate=0, **geo))
    # round 6: with a workspace the workgroups' partial rows are summed by a second launch in a fixed order (no atomics): same result,
    # accumulated INTO dw, and the same bits twice in a row
    dws = []
    for _ in range(2):
        d2 = DwBwdDesc(x=P(xd), dz=P(dzd), dw=None, lddx=0, accumulate=0, **geo)
        need = int(lib.yh_dw_wgrad_workspace(C.byref(d2)))
        assert DRY or need > 0
        ws = torch.full((max(need, 1),), float('nan'), device=GPU)
        dw2 = torch.ones(c, 1, k, k, device=GPU)
        d2.dw, d2.ws, d2.ws_floats = P(dw2), P(ws), need
        oh.call(lib, 'yh_dw_wgrad', d2)
        _sync()
        dws.append(dw2.clone())
    dx = torch.full((N, H, W, c), 3.0, device=GPU, dtype=dt)
    oh.call(lib, 'yh_dw_dgrad', DwBwdDesc(dz=P(dzd), w=P(packed), dx=P(dx), lddx=c, accumulate=0, **geo))
    dxa = acc0.to(GPU).clone()
    oh.call(lib, 'yh_dw_dgrad', DwBwdDesc(dz=P(dzd), w=P(packed), dx=P(dxa), lddx=c, accumulate=1, **geo))
    _sync()
    xr, dzr = x.float().permute(0, 3, 1, 2), dz.float().permute(0, 3, 1, 2)
    wq = w.to(dt).float()
    ref_w = torch.nn.grad.conv2d_weight(xr, (c, 1, k, k), dzr, stride=s, padding=pad, groups=c)
    ref_x = torch.nn.grad.conv2d_input((N, c, H, W), wq, dzr, stride=s, padding=pad, groups=c).permute(0, 2, 3, 1)
    assert (dw.cpu() - ref_w).abs().max().item() <= (2e-5 if code == F32 else 1e-4) * ref_w.abs().max().item()
    assert torch.equal(dws[0], dws[1])
    assert (dws[0].cpu() - 1.0 - ref_w).abs().max().item() <= (2e-5 if code == F32 else 1e-4) * ref_w.abs().max().item() + 1e-6
    tol = 2e-5 if code == F32 else 2.5e-3
    assert (dx.float().cpu() - ref_x).abs().max().item() <= tol * ref_x.abs().max().item()
    ref_a = ref_x + acc0.float()
    assert (dxa.float().cpu() - ref_a).abs().max().item() <= tol * ref_a.abs().max().item()


@pytest.mark.parametrize('code', [F32, F16], ids=['fp32', 'fp16'])
@pytest.mark.parametrize('case', [(3, 13, 13, 72), (2, 7, 9, 960), (4, 26, 26, 40)], ids=lambda c: 'n%d_%dx%d_c%d' % c)
def test_squeeze_excite_backward_matches_autograd(libs, code, case):
    from engine.hiplib import SeBwdDesc, SeDesc
    lib, _ = libs
    N, H, W, c = case
    cr = c // 4
    g = torch.Generator().manual_seed(c)
    dt = oh.tdtype(code)
    x = torch.randn(N, H, W, c, generator=g).to(dt)
    dy = torch.randn(N, H, W, c, generator=g).to(dt)
    w1 = (torch.randn(cr, c, generator=g) * c ** -0.5).contiguous()
    w2 = (torch.randn(c, cr, generator=g) * cr ** -0.5 * 3).contiguous()
    xd, dyd, w1d, w2d = x.to(GPU), dy.to(GPU), w1.to(GPU), w2.to(GPU)
    f32 = lambda *shape: torch.zeros(*shape, device=GPU, dtype=torch.float32)
    pooled, gate, scratch, dw1, dw2 = f32(N, c), f32(N, c), f32(N, c), f32(cr, c), f32(c, cr)
    y = torch.empty(N, H, W, c, device=GPU, dtype=dt)
    oh.call(lib, 'yh_se_fwd', SeDesc(x=P(xd), y=P(y), w1=P(w1d), w2=P(w2d), pooled=P(pooled), gate=P(gate), ch_map=None, n=N, h=H,
                                     w_in=W, c=c, c_phys=c, cr=cr, ldx=c, ldy=c, dtype=code))
    dx = torch.full((N, H, W, c), 3.0, device=GPU, dtype=dt)
    oh.call(lib, 'yh_se_bwd', SeBwdDesc(x=P(xd), dy=P(dyd), dx=P(dx), w1=P(w1d), w2=P(w2d), pooled=P(pooled), gate=P(gate), dw1=P(dw1),
                                        dw2=P(dw2), scratch=P(scratch), n=N, h=H, w_in=W, c=c, cr=cr, ldx=c, lddy=c, lddx=c,
                                        accumulate=0, dtype=code))
    _sync()
    xr = x.float().requires_grad_()
    w1r, w2r = w1.clone().requires_grad_(), w2.clone().requires_grad_()
    gt = F.relu6(F.linear(F.relu(F.linear(xr.mean((1, 2)), w1r)), w2r) + 3.0) / 6.0
    (xr * gt.view(N, 1, 1, c)).backward(dy.float())
    tol = 5e-5 if code == F32 else 3e-3
    assert (dx.float().cpu() - xr.grad).abs().max().item() <= tol * xr.grad.abs().max().item()
    assert (dw1.cpu() - w1r.grad).abs().max().item() <= tol * (w1r.grad.abs().max().item() + 1e-6)
    assert (dw2.cpu() - w2r.grad).abs().max().item() <= tol * (w2r.grad.abs().max().item() + 1e-6)


# ------------------------------------------------------------------------------------------ whole steps
@pytest.fixture(scope='module')
def mini():
    path = th.write_cfg(th.mini_cfg_text())
    yield path
    os.unlink(path)


def _engine_lib():
    return fakelib.FakeLib() if DRY else None


@pytest.mark.parametrize('size', [64, 52])
def test_mini_train_step_fp32_matches_eager_autograd(libs, mini, size):
    model = th.build(mini, size)
    x = synth.image_batch(4, size, seed=0)
    raws_ref, grads_ref, m_ref, ws = th.eager_step(model, x)
    raws, grads, m = th.engine_step(model, x, ws, 'fp32', lib=_engine_lib(), device=GPU)
    for a, b in zip(raws, raws_ref):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
    for k in grads_ref:
        assert th.rel_l2(grads[k], grads_ref[k]) < 5e-5, k
    for (k, a), (_, b) in zip(m.state_dict().items(), m_ref.state_dict().items()):
        if 'running' in k:
            assert (a.cpu() - b).abs().max().item() <= 1e-5 * (b.abs().max().item() + 1), k


def test_single_channel_train_step_matches_eager(libs):
    """Gray-scale input (reference cfg/yolov3-singlechannel, is_gray_scale=True) through the stem and its MFMA weight gradient."""
    import test_train_emulated as tte
    model = tte._gray_model()
    x = torch.rand(4, 1, 64, 64, generator=torch.Generator().manual_seed(3))
    raws_ref, grads_ref, _, ws = th.eager_step(model, x)
    raws, grads, _ = th.engine_step(model, x, ws, 'fp32', lib=_engine_lib(), device=GPU)
    for a, b in zip(raws, raws_ref):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item()
    tte.assert_grads_close(grads, grads_ref, 5e-5)


def test_mini_train_step_fp16_close_to_emulated_fp16(libs, mini):
    """fp16 kernels vs the fp16 emulation of the same plan (same rounding points): statistical, kinks flip."""
    model = th.build(mini, 64)
    x = synth.image_batch(4, 64, seed=0)
    _, grads_ref, _, ws = th.eager_step(model, x)
    raws_e, grads_e, _ = th.engine_step(model, x, ws, 'fp16', lib=fakelib.FakeLib())
    raws, grads, _ = th.engine_step(model, x, ws, 'fp16', lib=_engine_lib(), device=GPU)
    for a, b in zip(raws, raws_e):
        assert (a - b).abs().max().item() <= 5e-3 * b.abs().max().item()
    for k in grads_ref:
        assert th.cosine(grads[k], grads_ref[k]) > 0.97, k
        assert th.cosine(grads[k], grads_e[k]) > 0.97, k


def test_mini_sgd_steps_track_eager(libs, mini):
    """Five SGD steps on the GPU path vs five eager steps on CPU: parameters stay together (fp32)."""
    model = th.build(mini, 64)
    import copy
    a = copy.deepcopy(model).train()
    b = copy.deepcopy(model).to(GPU).train()
    if DRY:
        from engine.train import TrainEngine
        b.__dict__['_hip_train_engine'] = TrainEngine(b, 'fp32', lib=fakelib.FakeLib())
    oa = torch.optim.SGD(a.parameters(), lr=2e-6, momentum=0.9)
    ob = torch.optim.SGD(b.parameters(), lr=2e-6, momentum=0.9)
    ws = None
    for step in range(5):
        x = synth.image_batch(4, 64, seed=step)
        ra = a._forward_eager(x)[0]
        ws = ws or th.loss_weights(ra)
        oa.zero_grad()
        th.toy_loss(ra, ws).backward()
        oa.step()
        rb = (b._forward_hip_train(x.to(GPU)) if DRY else b(x.to(GPU)))[0]
        ob.zero_grad()
        th.toy_loss(rb, ws).backward()
        ob.step()
    for (k, pa), (_, pb) in zip(a.state_dict().items(), b.state_dict().items()):
        if pa.dtype.is_floating_point:
            # 3e-4: the wgrad split sums meet in fp32 atomics whose order varies run to run; over five steps one leaky-ReLU kink that
            # flips on that last-bit difference moves a weight by ~1e-4 relative (seen once in ~5 runs at 1e-4)
            assert (pa - pb.cpu()).abs().max().item() <= 3e-4 * (pa.abs().max().item() + 1e-3), k


@pytest.mark.parametrize('rel', ['yolov3tiny/yolov3-tiny.cfg', 'yolov4/yolov4.cfg', 'yolov4tiny/yolov4-tiny.cfg',
                                 'yolov3-mobilenet/yolov3-mobilenet-coco.cfg'], ids=['yolov3-tiny', 'yolov4', 'yolov4-tiny', 'mobilenet'])
def test_maxpool_graphs_train_step_against_fp64(libs, rel):
    """yolov3-tiny (maxpools) and YOLOv4 (mish, SPP, PAN) on the GPU training path vs an fp64 eager run.

    The two tiny nets end in 4 x 4 maps at this input size (64 values per channel and batch): ONE leaky-ReLU sign or max-pool
    arg-max that falls on the other side of a tie moves every gradient upstream of it by 1e-3 .. 2e-2 of the total norm.  Which
    side a tie falls on depends on the last bit of the first layer's batch statistics: scaling the first BatchNorm's variance by
    1 +- 1e-6 in the separate-statistics path (round 4, a debug build) fails the 2e-4 bound exactly like the statistics epilogue
    of the MFMA first-layer kernel does (5.8e-4 / 1.7e-2, every parameter above the flipped unit off by the same relative
    amount, the heads and the forward outputs equal to 2e-6).  The forward bound stays tight; the gradient bound of the two
    tiny nets allows for such a flip.  The flip-free evidence for the epilogue is test_stem_on_the_matrix_cores_with_statistics
    (its sums are the sums of the stored values) and the first-block tests of this file."""
    cfg = os.path.join(conftest.PKG, 'cfg', rel)
    model = th.build(cfg, 128)
    x = synth.image_batch(4, 128, seed=0)
    raws64, grads64, _, ws = th.eager_step(model, x, dtype=torch.float64)
    _, grads32, _, _ = th.eager_step(model, x, ws=ws)
    raws, grads, _ = th.engine_step(model, x, ws, 'fp32', lib=_engine_lib(), device=GPU)
    for a, b in zip(raws, raws64):
        assert (a - b).abs().max().item() <= 3e-4 * b.abs().max().item()
    num = sum((grads[k] - grads64[k]).norm().item() ** 2 for k in grads64) ** 0.5
    den = sum((grads32[k] - grads64[k]).norm().item() ** 2 for k in grads64) ** 0.5
    tot = sum(grads64[k].norm().item() ** 2 for k in grads64) ** 0.5
    flip = {'yolov3tiny/yolov3-tiny.cfg': 3e-3, 'yolov4tiny/yolov4-tiny.cfg': 5e-2}.get(rel, 2e-4)
    assert num <= 4 * den + flip * tot, (num / tot, den / tot)


def test_yolov3_train_step_against_fp64(libs):
    cfg = os.path.join(conftest.PKG, 'cfg', 'yolov3', 'yolov3.cfg')
    model = th.build(cfg, 128)
    x = synth.image_batch(4, 128, seed=0)
    raws64, grads64, _, ws = th.eager_step(model, x, dtype=torch.float64)
    _, grads32, _, _ = th.eager_step(model, x, ws=ws)
    raws, grads, _ = th.engine_step(model, x, ws, 'fp32', lib=_engine_lib(), device=GPU)
    for a, b in zip(raws, raws64):
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item()
    num = sum((grads[k] - grads64[k]).norm().item() ** 2 for k in grads64) ** 0.5
    den = sum((grads32[k] - grads64[k]).norm().item() ** 2 for k in grads64) ** 0.5
    tot = sum(grads64[k].norm().item() ** 2 for k in grads64) ** 0.5
    assert num <= 4 * den + 1e-4 * tot, (num / tot, den / tot)


def _protocol_targets(batch, seed=1):
    """nt = 8 N rows [img, cls, x, y, w, h], xy ~ U(0.1, 0.9), wh ~ U(0.05, 0.35), seed 1."""
    g = torch.Generator().manual_seed(seed)
    n = 8 * batch
    img = torch.arange(batch).repeat_interleave(8).float().view(-1, 1)
    cls = torch.randint(0, 80, (n, 1), generator=g).float()
    xy = torch.rand(n, 2, generator=g) * 0.8 + 0.1
    wh = torch.rand(n, 2, generator=g) * 0.30 + 0.05
    return torch.cat((img, cls, xy, wh), 1)


@pytest.mark.parametrize('precision,loss_tol,norm_tol', [('fp32', 1e-3, 1e-2), ('fp16', 1e-2, 5e-2)], ids=['fp32', 'fp16'])
def test_training_step_protocol_loss_and_grad_norm(libs, precision, loss_tol, norm_tol):
    """The SURVEY 8(d) training-parity protocol on YOLOv3 @320, batch 2: train-mode forward + compute_loss (train.py's
    default hyp, gr = 1) + backward on the GPU path against the same step on the CPU eager modules (fp32): loss items
    within 1e-3 relative and the global gradient norm within 1e-2 relative (fp16 engine vs the fp32 oracle: 1e-2 / 5e-2)."""
    import copy
    from models import Darknet
    from utils.utils import compute_loss
    hyp = {'giou': 3.54, 'cls': 37.4, 'cls_pw': 1.0, 'obj': 64.3, 'obj_pw': 1.0, 'iou_t': 0.20, 'fl_gamma': 0.0}
    torch.manual_seed(0)
    model = Darknet(os.path.join(conftest.PKG, 'cfg', 'yolov3', 'yolov3.cfg'), (320, 320)).train()
    state = model.state_dict()
    synth.randomize_bn_(state, seed=1)
    model.load_state_dict(state)
    model.nc, model.hyp, model.gr = 80, hyp, 1.0
    x = synth.image_batch(2, 320, seed=0)
    targets = _protocol_targets(2)
    ref = copy.deepcopy(model)
    pred, _ = ref(x)
    loss_ref, items_ref = compute_loss(pred, targets, ref)
    loss_ref.backward()
    norm_ref = sum(p.grad.norm().item() ** 2 for p in ref.parameters()) ** 0.5
    dev = copy.deepcopy(model).to(GPU)
    os.environ['YOLO_HIP_TRAIN_PRECISION'] = precision
    try:
        if DRY:
            from engine.train import TrainEngine
            dev.__dict__['_hip_train_engine'] = TrainEngine(dev, precision, lib=fakelib.FakeLib())
            pred, _ = dev._forward_hip_train(x)
        else:
            pred, _ = dev(x.to(GPU))
        loss, items = compute_loss(pred, targets.to(GPU), dev)
        loss.backward()
    finally:
        del os.environ['YOLO_HIP_TRAIN_PRECISION']
    assert dev.__dict__.get('_hip_train_engine') is not None
    rel = ((items.cpu() - items_ref).abs() / items_ref.abs().clamp(min=1e-6)).max().item()
    assert rel <= loss_tol, (items.cpu(), items_ref)
    norm = sum(p.grad.norm().item() ** 2 for p in dev.parameters()) ** 0.5
    assert abs(norm - norm_ref) <= norm_tol * norm_ref, (norm, norm_ref)


def test_headline_shape_fp16_step_against_the_fp32_engine(libs):
    """The bench's headline shape itself — YOLOv3-608, batch 64, fp16 (autocast recipe) — against the fp32 engine on the same
    inputs and weights (the fp32 engine is pinned to eager autograd / the reference goldens by the tests above; an eager CPU step
    at this size takes minutes).  SURVEY 8(d) fp16 tolerances: loss items 1e-2 relative, global gradient norm 5e-2 relative;
    additionally every head tensor's drift is bounded and the gradient DIRECTION is as close as the conditioning of the problem
    allows.  Through 75 random-weight convs a leaky-ReLU kink that flips under a 1e-3 relative perturbation moves every upstream
    gradient, so the yardstick is measured in the same test: the fp32 engine run once more with only the WEIGHTS and the input
    rounded to fp16 (activations, gradients and accumulation still fp32).  The fp16 engine, which also rounds every activation and
    gradient tensor, may lose at most 4x that run's gradient-direction error (1 - cosine)."""
    if DRY:
        pytest.skip('needs the GPU (27 TFLOP per step)')
    import copy
    from models import Darknet
    from utils.utils import compute_loss
    hyp = {'giou': 3.54, 'cls': 37.4, 'cls_pw': 1.0, 'obj': 64.3, 'obj_pw': 1.0, 'iou_t': 0.20, 'fl_gamma': 0.0}
    torch.manual_seed(0)
    model = Darknet(os.path.join(conftest.PKG, 'cfg', 'yolov3', 'yolov3.cfg'), (608, 608)).train()
    model.nc, model.hyp, model.gr = 80, hyp, 1.0
    batch = 64
    x = synth.image_batch(batch, 608, seed=3).to(GPU)
    targets = _protocol_targets(batch).to(GPU)
    out = {}
    for tag in ('fp32', 'fp16', 'fp32_rounded_inputs'):
        precision = tag[:4]
        dev = copy.deepcopy(model).to(GPU)
        xin = x
        if tag == 'fp32_rounded_inputs':
            with torch.no_grad():
                for prm in dev.parameters():
                    prm.copy_(prm.half().float())
            xin = x.half().float()
        os.environ['YOLO_HIP_TRAIN_PRECISION'] = precision
        try:
            pred, _ = dev(xin)
            loss, items = compute_loss(pred, targets, dev)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            del os.environ['YOLO_HIP_TRAIN_PRECISION']
        eng = dev.__dict__.get('_hip_train_engine')
        assert eng is not None and eng.precision == precision
        out[tag] = dict(items=items.cpu(), pred=[p.detach().float().cpu() for p in pred],
                              grads={k: p.grad.detach().cpu() for k, p in dev.named_parameters()})
        dev.__dict__['_hip_train_engine'] = None
        del dev, eng, pred, loss
        torch.cuda.empty_cache()
    a, b = out['fp16'], out['fp32']
    rel = ((a['items'] - b['items']).abs() / b['items'].abs().clamp(min=1e-6)).max().item()
    na = sum(g.norm().item() ** 2 for g in a['grads'].values()) ** 0.5
    nb = sum(g.norm().item() ** 2 for g in b['grads'].values()) ** 0.5
    drift = max((p - q).abs().max().item() / q.abs().max().item() for p, q in zip(a['pred'], b['pred']))
    energy = sorted(((g.norm().item() ** 2, k) for k, g in b['grads'].items()), reverse=True)
    acc, top = 0.0, []
    for e, k in energy:
        top.append(k)
        acc += e
        if acc >= 0.99 * nb ** 2:
            break
    cos_min = min(torch.nn.functional.cosine_similarity(a['grads'][k].flatten(), b['grads'][k].flatten(), dim=0).item() for k in top)
    def cosine(u, v):
        dot = sum((u[k].double() * v[k].double()).sum().item() for k in v)
        return dot / (sum(g.double().norm().item() ** 2 for g in u.values()) ** 0.5 * sum(g.double().norm().item() ** 2 for g in v.values()) ** 0.5)
    cos = cosine(a['grads'], b['grads'])
    cos_yard = cosine(out['fp32_rounded_inputs']['grads'], b['grads'])
    print('608 b64 fp16 vs fp32 engine: loss items rel %.3g, grad norm %.6g vs %.6g (rel %.3g), head drift %.3g, gradient cosine %.4f '
          '(weakest of %d dominant parameters %.4f); fp32 engine with fp16-rounded weights + input: cosine %.4f'
          % (rel, na, nb, abs(na - nb) / nb, drift, cos, len(top), cos_min, cos_yard))
    assert rel <= 1e-2, (a['items'], b['items'])
    assert abs(na - nb) <= 5e-2 * nb, (na, nb)
    assert drift <= 2e-2
    assert 1.0 - cos <= 4.0 * (1.0 - cos_yard) + 0.01, (cos, cos_yard)
    # the relative bound above would pass at cosine 0.77; measured on this ill-conditioned default-init net: 0.9176
    # (yardstick 0.9439).  Twice the measured direction error is the absolute ceiling.
    assert 1.0 - cos <= 2.0 * (1.0 - 0.9176), cos


@pytest.mark.parametrize('tag,rel,size,nc', [('tinyhand', 'yolov3tiny/yolov3-tiny-hand.cfg', 128, 1), ('v4tiny', 'yolov4tiny/yolov4-tiny.cfg', 128, 80)],
                         ids=['tinyhand', 'v4tiny'])
def test_training_step_on_gpu_matches_reference_golden(libs, tag, rel, size, nc):
    """The reference's own training step (tests/golden/train_step.npz, generated from /root/reference on CPU) vs the GPU
    path: HIP train forward + fused compute_loss + HIP backward, fp32.  Loss items 1e-4; gradients in the l2 sense
    (isolated maxpool-argmax / leaky-kink flips move single elements)."""
    if DRY:
        pytest.skip('covered on the emulator by tests/test_train_emulated.py')
    import test_train_emulated as tte
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'train_step.npz'))
    model, pred, items = tte._golden_step(tag, rel, size, nc, path='gpu', device=GPU)
    assert model.__dict__.get('_hip_train_engine') is not None
    np.testing.assert_allclose(items.detach().cpu().numpy(), gold[tag + '_items'], rtol=1e-4)
    params = dict(model.named_parameters())
    num = den = 0.0
    for k, want in zip([str(n) for n in gold[tag + '_param_names']], gold[tag + '_grad_checks']):
        rows = gold['%s_grow_%s' % (tag, k)]
        got = params[k].grad.reshape(-1)[::211].cpu().numpy()
        num += float(((got - rows) ** 2).sum())
        den += float((rows ** 2).sum())
        assert abs(params[k].grad.abs().sum().item() - want[1]) <= 1e-2 * want[1] + 1e-6, k
    assert (num / den) ** 0.5 <= 3e-3


def _worst_parameters(rows, want, lens, names, top=6):
    """Which parameters carry the difference: (name, share of the squared error, own rel l2) of the `top` largest contributors."""
    out, at, tot = [], 0, float(((rows - want) ** 2).sum()) + 1e-300
    for k, n in zip(names, lens):
        d, w = rows[at:at + n] - want[at:at + n], want[at:at + n]
        out.append((k, float((d ** 2).sum()) / tot, float(np.linalg.norm(d) / (np.linalg.norm(w) + 1e-30))))
        at += int(n)
    return sorted(out, key=lambda t: -t[1])[:top]


def _reference_golden_step(gold_file, cfg_rel, calm=False):
    """(gold, step): `step(precision, round_inputs)` runs ONE training step of this package's Darknet(cfg_rel) on the GPU in the state
    and on the inputs the golden generator gave the reference (tests/golden/make_golden_train608.py / make_golden_train_v4.py) and
    returns loss items, raw heads, sampled gradient rows with their rel l2 / cosine / norm ratio against the golden's, per-parameter
    |grad| sums, the l2 norm of the whole gradient and the state_dict.  round_inputs: weights and frames rounded to fp16 first (the
    conditioning yardstick: what ANY fp16 engine starts from)."""
    import copy
    from models import Darknet
    from utils.utils import compute_loss
    import test_train_emulated as tte
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', gold_file))
    size, batch, stride = int(gold['size']), int(gold['batch']), int(gold['stride'])
    torch.manual_seed(0)
    model0 = Darknet(os.path.join(conftest.PKG, 'cfg', *cfg_rel.split('/')), (size, size))
    state = model0.state_dict()
    synth.randomize_bn_(state, seed=1)
    if calm:
        synth.calm_bn_(state)
    model0.load_state_dict(state)
    model0.nc, model0.hyp, model0.gr = 80, dict(tte.GOLD_HYP), 1.0
    targets = synth.loss_inputs(model0, size, batch=batch, seed=9, labels_per_image=8)[1].to(GPU)
    x0 = synth.image_batch(batch, size, seed=0).to(GPU)
    names = [str(n) for n in gold['param_names']]
    want = gold['grad_rows']

    def step(prec, round_inputs=False):
        model = copy.deepcopy(model0)
        x = x0
        if round_inputs:
            with torch.no_grad():
                for prm in model.parameters():
                    prm.copy_(prm.half().float())
            x = x0.half().float()
        model.train().to(GPU)
        os.environ['YOLO_HIP_TRAIN_PRECISION'] = prec
        try:
            pred, _ = model(x)
        finally:
            del os.environ['YOLO_HIP_TRAIN_PRECISION']
        eng = model.__dict__.get('_hip_train_engine')
        assert eng is not None and eng.precision == prec
        loss, items = compute_loss(pred, targets, model)
        gscale = 4096.0 if prec == 'fp16' else 1.0       # GradScaler: unscaled, the early layers' batch-2 gradients are fp16 subnormals
        (loss * gscale).backward()
        params = dict(model.named_parameters())
        assert names == list(params)
        for prm in params.values():
            prm.grad.div_(gscale)
        rows = np.concatenate([params[k].grad.reshape(-1)[::stride].float().cpu().numpy() for k in names])
        assert rows.shape == want.shape
        out = dict(items=items.detach().cpu().numpy(), pred=[p.detach().double().cpu() for p in pred], rows=rows,
                   sums=np.array([params[k].grad.abs().sum().item() for k in names]),
                   gnorm=float(sum(params[k].grad.double().pow(2).sum().item() for k in names) ** 0.5), sd={k: v.double().cpu() for k, v in model.state_dict().items()},
                   rel=float(np.linalg.norm(rows - want) / np.linalg.norm(want)), worst=_worst_parameters(rows, want, gold['grad_rows_len'], names),
                   cos=float((rows * want).sum() / (np.linalg.norm(rows) * np.linalg.norm(want))), nrm=float(np.linalg.norm(rows) / np.linalg.norm(want)))
        model.__dict__['_hip_train_engine'] = None
        del model, eng, pred, loss
        torch.cuda.empty_cache()
        return out

    return gold, step


# fp16 engine vs the reference-generated 608 golden, measured when the case was added (profiles/r05_pytest_gpu_tail.txt, two runs): gradient
# rel l2 0.378 / 0.436, cosine 0.927 / 0.900 - with the yardstick (fp32 engine, fp16-rounded weights + input) at rel l2 0.268, cosine 0.965:
# absolute caps = 1.5 x the worse rel l2 and twice the worse direction error
FP16_GOLD608_REL, FP16_GOLD608_COS = 0.65, 0.80


@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
def test_headline_shape_fp32_step_against_the_reference_golden(libs, precision):
    """the external anchor AT the shape the bench times.  tests/golden/train_step_608.npz is one training step of
    the REFERENCE itself (YOLOv3 Darknet-53, 608 x 608, batch 2, fp32 CPU: train-mode forward, compute_loss, backward; generated by
    tests/golden/make_golden_train608.py from /root/reference).  The fp32 HIP step on the GPU (train forward + fused compute_loss +
    backward, every kernel of the fp32 path at 608 geometry) against it: loss items, raw-head checksums, running statistics, and the
    parameter gradients in the l2 / cosine sense - a 75-conv random-weight net is ill-conditioned (one leaky-ReLU kink that flips
    under a different summation order moves every upstream gradient: eager fp32 itself is ~1e-2 from an fp64 run).
    precision = fp16: the engine the bench times - fp16 activations / gradients, fp32 master weights, the loss
    scaled as train.py's GradScaler does - against the SAME reference-generated step.  Loss items, head checksums and running
    statistics to SURVEY 8(d)'s 1e-2.  The gradient DIRECTION cannot be held to 1e-2 on this net by any fp16 engine, and the test
    measures why: the fp32 engine, with nothing but its weights and input rounded to fp16 (activations, gradients, accumulation all
    fp32), is run against the same golden as the yardstick (measured: rel l2 0.27, cosine 0.965 - a 5e-4 relative perturbation of the
    weights alone turns the sampled gradient by 15 degrees); the fp16 engine may lose at most three times the yardstick's direction
    error and twice its rel l2, and is capped at the absolute values above."""
    if DRY:
        pytest.skip('a 608 x 608 Darknet-53 step on the host emulation takes minutes; the emulator is covered at 64 - 128 px')
    gold, step = _reference_golden_step('train_step_608.npz', 'yolov3/yolov3.cfg')
    f16 = precision == 'fp16'
    r = step(precision)
    np.testing.assert_allclose(r['items'], gold['items'], rtol=1e-2 if f16 else 2e-3)
    for i, got in enumerate(r['pred']):
        w = gold['raw%d_checks' % i]
        assert abs(got.abs().sum().item() - w[1]) <= (1e-2 if f16 else 2e-3) * w[1] and abs(got.abs().max().item() - w[2]) <= (3e-2 if f16 else 1e-2) * w[2], i
    ratio = r['sums'] / np.maximum(gold['grad_checks'][:, 1], 1e-30)
    big = gold['grad_checks'][:, 1] >= 1e-3 * gold['grad_checks'][:, 1].max()
    print('608 b2 %s HIP step vs the reference golden: loss items %s, gradient rel l2 %.3g, cosine %.5f, norm ratio %.4f, |grad| sum ratio '
          'of the dominant parameters %.3f .. %.3f' % (precision, r['items'], r['rel'], r['cos'], r['nrm'], ratio[big].min(), ratio[big].max()))
    if f16:
        y = step('fp32', round_inputs=True)
        print('608 b2 yardstick (fp32 engine, weights + input rounded to fp16) vs the reference golden: gradient rel l2 %.3g, cosine %.5f, '
              'norm ratio %.4f' % (y['rel'], y['cos'], y['nrm']))
        assert 1.0 - r['cos'] <= 3.0 * (1.0 - y['cos']) + 0.01 and r['rel'] <= 2.0 * y['rel'] + 0.02, (r['rel'], r['cos'], y['rel'], y['cos'])
        assert r['rel'] <= FP16_GOLD608_REL and r['cos'] >= FP16_GOLD608_COS, (r['rel'], r['cos'])
        assert 0.8 <= ratio[big].min() and ratio[big].max() <= 1.25
    else:
        assert r['rel'] <= 0.1 and r['cos'] >= 0.995, (r['rel'], r['cos'])
        assert 0.8 <= ratio[big].min() and ratio[big].max() <= 1.25
    for k, w in zip([str(n) for n in gold['running_names']], gold['running_checks']):
        assert abs(r['sd'][k].abs().sum().item() - w[1]) <= (3e-3 if f16 else 1e-3) * w[1] + 1e-6, k


@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
@pytest.mark.parametrize('net', ['v3', 'v4'])
def test_well_conditioned_step_against_the_reference_golden(libs, net, precision):
    """a WELL-CONDITIONED external anchor for the training step at the bench shape.  tests/golden/train_step_<net>_608_calm.npz
    is one training step of the REFERENCE itself - YOLOv3 (the network the bench times) and YOLOv4 (cfg/yolov4/yolov4.cfg: CSPDarknet53 with
    Mish, SPP, PAN) at 608 x 608, batch 2, fp32 CPU, tests/golden/make_golden_train_calm.py imports /root/reference - in the state of
    synth.calm_bn_: small BatchNorm gains, so that perturbations are not amplified through the depth of the net.  (In the plain random
    state rounding the weights to fp16 alone moves Darknet-53's gradient by rel l2 0.34 and YOLOv4's by 1.5 - the verdict's hope that a
    Mish net would be smooth did not hold, the amplification belongs to deep random batch-norm nets, not to the kinks - and the test
    above can only cap the fp16 engine loosely.)  Here SURVEY 8(d)'s numbers are asserted AS WRITTEN for both engines: loss items 1e-2
    (fp32: 1e-4), l2 norm of the whole gradient 1e-2 (fp32: 1e-3); and the gradient itself: rel l2 of the sampled rows <= 2e-3 for the
    fp32 engine; for the fp16 engine <= 5e-2 absolute and <= 2 x the REFERENCE'S OWN mixed-precision step - the golden also holds the
    reference run under torch.autocast(float16) (its --mpt recipe, train.py:371-374, on the CPU) against its fp32 step: rel l2 0.024 on
    Darknet-53.  fp16 storage of activations and gradients costs more than fp16 weights (printed as the second yardstick: 5e-3): the
    per-channel BatchNorm gradients are sums with heavy cancellation, and they carry most of the difference on both sides.
    Measured values: DESIGN.md section 4 / profiles/r06_pytest_gpu_tail_final.txt."""
    if DRY:
        pytest.skip('a 608 x 608 step on the host emulation takes minutes; the emulator is covered at 64 - 128 px')
    gold, step = _reference_golden_step('train_step_%s_608_calm.npz' % net, {'v3': 'yolov3/yolov3.cfg', 'v4': 'yolov4/yolov4.cfg'}[net], calm=True)
    f16 = precision == 'fp16'
    r = step(precision)
    ratio = r['sums'] / np.maximum(gold['grad_checks'][:, 1], 1e-30)
    big = gold['grad_checks'][:, 1] >= 1e-3 * gold['grad_checks'][:, 1].max()
    gn = r['gnorm'] / float(gold['grad_norm'])
    print('calm %s 608 b2 %s HIP step vs the reference golden: loss items %s (reference %s), gradient norm ratio %.5f, sampled rows rel l2 %.3g, '
          'cosine %.6f, |grad| sum ratio of the dominant parameters %.4f .. %.4f' % (net, precision, r['items'], gold['items'], gn, r['rel'], r['cos'],
                                                                                    ratio[big].min(), ratio[big].max()))
    print('   largest contributors to the difference (parameter, share of the squared error, own rel l2): %s' % ', '.join('%s %.2f %.2g' % w for w in r['worst']))
    y = None
    if f16:
        y = step('fp32', round_inputs=True)
        print('calm %s 608 b2 yardstick (fp32 engine, weights + input rounded to fp16) vs the reference golden: gradient norm ratio %.5f, rel l2 %.3g, '
              'cosine %.6f' % (net, y['gnorm'] / float(gold['grad_norm']), y['rel'], y['cos']))
    np.testing.assert_allclose(r['items'], gold['items'], rtol=1e-2 if f16 else 1e-4)
    for i, got in enumerate(r['pred']):
        w = gold['raw%d_checks' % i]
        assert abs(got.abs().sum().item() - w[1]) <= (1e-2 if f16 else 1e-4) * w[1], i
    assert abs(gn - 1.0) <= (1e-2 if f16 else 1e-3), gn                 # gradient norm 1e-2
    assert r['rel'] <= 5e-2, r['rel']
    if f16:
        mpt = float(gold['mpt_rel'])
        print('calm %s 608 b2: the reference under torch.autocast(float16) against its own fp32 step: sampled rows rel l2 %.3g, gradient norm ratio %.5f'
              % (net, mpt, float(gold['mpt_grad_norm']) / float(gold['grad_norm'])))
        assert r['rel'] <= 2.0 * mpt, (r['rel'], mpt)
    else:
        assert r['rel'] <= 2e-3 and r['cos'] >= 0.99999, (r['rel'], r['cos'])
    # per-parameter |grad| sums of the dominant parameters (measured: fp32 0.9990 .. 1.0005, fp16 0.963 .. 1.012)
    assert (0.95 if f16 else 0.995) <= ratio[big].min() and ratio[big].max() <= (1.05 if f16 else 1.005)
    for k, w in zip([str(n) for n in gold['running_names']], gold['running_checks']):
        assert abs(r['sd'][k].abs().sum().item() - w[1]) <= (3e-3 if f16 else 1e-4) * w[1] + 1e-6, k


@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
def test_sgd_trajectory_against_the_reference_golden(libs, precision):
    """what "training parity" means to a user - the first ten optimisation steps of the REFERENCE's loop at the
    headline shape (YOLOv3-608, batch 2; train.py:344-455 with its burn-in schedule and three-group nesterov SGD, stated once in
    tests/sgd_protocol.py; tests/golden/sgd_608.npz generated from /root/reference's own Darknet + compute_loss on the CPU, in the
    well-conditioned state of synth.calm_bn_ - in the plain random state the fp32 engine itself is 11 % off by step ten) against
    the same loop on this package's Darknet on the GPU: the fp32 engine, and the fp16 engine as train.py --mpt drives it (autocast +
    GradScaler).  Per step: the four loss items within 1e-2 (fp32: 1e-3), the parameter norm within 1e-5, the displacement from the
    initial point within 1e-2 (fp32: 2e-3); at the end the parameters and running statistics by checksum."""
    if DRY:
        pytest.skip('ten 608 x 608 Darknet-53 steps on the host emulation take an hour')
    import sgd_protocol
    from models import Darknet
    from utils.utils import compute_loss
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sgd_608.npz'))
    size, batch, steps = int(gold['size']), int(gold['batch']), int(gold['steps'])
    torch.manual_seed(0)
    model = Darknet(os.path.join(conftest.PKG, 'cfg', 'yolov3', 'yolov3.cfg'), (size, size))
    state = model.state_dict()
    synth.calm_bn_(synth.randomize_bn_(state, seed=1))
    model.load_state_dict(state)
    model.to(GPU)
    f16 = precision == 'fp16'
    tr = sgd_protocol.run(model, compute_loss, steps, size, batch, device=GPU, mixed=f16)
    eng = model.__dict__.get('_hip_train_engine')
    assert eng is not None and eng.precision == precision
    assert np.array_equal(tr['stepped'], gold['stepped'])
    ei = np.abs(tr['items'] / gold['items'] - 1.0).max(1)
    ed = np.abs(tr['dnorm'] / gold['dnorm'] - 1.0)
    ep = np.abs(tr['pnorm'] / gold['pnorm'] - 1.0)
    print('sgd trajectory 608 b2 %s vs the reference: per step max loss-item error %s; displacement-norm error %s; parameter-norm error <= %.2g'
          % (precision, ' '.join('%.1e' % v for v in ei), ' '.join('%.1e' % v for v in ed), ep.max()))
    assert ei.max() <= (1e-2 if f16 else 1e-3), ei
    assert ed.max() <= (1e-2 if f16 else 2e-3), ed
    assert ep.max() <= 1e-5, ep
    sd = model.state_dict()
    worst = 0.0
    for k, w in zip([str(n) for n in gold['state_names']], gold['state_checks']):
        got = sd[k].double().abs().sum().item()
        worst = max(worst, abs(got - w[1]) / (w[1] + 1e-6))
        assert abs(got - w[1]) <= (5e-3 if f16 else 1e-3) * w[1] + 1e-6, k
    print('sgd trajectory 608 b2 %s: largest relative |.|-sum error over %d final tensors (parameters + running statistics) %.2e' % (precision, len(gold['state_names']), worst))


@pytest.mark.parametrize('net,precision', [('yolov3/yolov3.cfg', 'fp16'), ('yolov3/yolov3.cfg', 'fp32'), ('yolov4/yolov4.cfg', 'fp16'),
                                           ('yolov3-mobilenet/yolov3-mobilenet-coco.cfg', 'fp16')],
                         ids=['yolov3-fp16', 'yolov3-fp32', 'yolov4-fp16', 'mobilenet-fp16'])
def test_training_step_is_run_to_run_deterministic(libs, net, precision):
    """the reference's CPU training step is bit-reproducible; so is this one.  YOLOv3 at 320 x 320, batch 4: the same
    step (train-mode forward, fused compute_loss, backward) three times from the same state - `torch.equal` on every parameter gradient,
    on the raw heads and on the running statistics.  Every sum over workgroups on this path is taken in a fixed order (common.h
    deterministic(): BatchNorm statistics and backward sums, weight-gradient pixel splits, the first block's backward).  Later in round 6
    also YOLOv4 (SPP max-pool backward as a gather instead of fp16 atomics) and YOLOv3-Mobilenetv3 (depthwise and squeeze-excite weight
    gradients as ordered partial rows).  Not covered: three or more labels in one (cell, anchor), the loss items' sums."""
    if DRY:
        pytest.skip('the host emulation is sequential: nothing to show')
    import copy
    from models import Darknet
    from utils.utils import compute_loss
    import test_train_emulated as tte
    size, batch = 320, 4
    torch.manual_seed(0)
    model0 = Darknet(os.path.join(conftest.PKG, 'cfg', *net.split('/')), (size, size))
    state = model0.state_dict()
    synth.randomize_bn_(state, seed=1)
    model0.load_state_dict(state)
    model0.nc, model0.hyp, model0.gr = 80, dict(tte.GOLD_HYP), 1.0
    targets = synth.loss_inputs(model0, size, batch=batch, seed=9, labels_per_image=8)[1].to(GPU)
    x = synth.image_batch(batch, size, seed=0).to(GPU)

    def step():
        model = copy.deepcopy(model0).train().to(GPU)
        os.environ['YOLO_HIP_TRAIN_PRECISION'] = precision
        try:
            pred, _ = model(x)
        finally:
            del os.environ['YOLO_HIP_TRAIN_PRECISION']
        assert model.__dict__.get('_hip_train_engine') is not None
        loss, items = compute_loss(pred, targets, model)
        (loss * (1024.0 if precision == 'fp16' else 1.0)).backward()
        torch.cuda.synchronize()
        out = ({k: p.grad.detach().clone() for k, p in model.named_parameters()}, [p.detach().clone() for p in pred],
               {k: v.detach().clone() for k, v in model.state_dict().items() if 'running' in k})
        model.__dict__['_hip_train_engine'] = None
        return out
    first = step()
    for run in (1, 2):
        again = step()
        diff = [k for k in first[0] if not torch.equal(first[0][k], again[0][k])]
        assert not diff, 'run %d: %d of %d parameter gradients differ from run 0, e.g. %s' % (run, len(diff), len(first[0]), diff[:3])
        assert all(torch.equal(a, b) for a, b in zip(first[1], again[1])), 'raw heads differ'
        assert all(torch.equal(first[2][k], again[2][k]) for k in first[2]), 'running statistics differ'
    print('%s 320 b4 %s: three runs of the training step, %d parameter gradients + 3 raw heads + %d running statistics bit-identical'
          % (net.split('/')[0], precision, len(first[0]), len(first[2])))


def test_step_with_backward_sums_in_the_data_gradient_equals_the_plain_step(libs, monkeypatch):
    """Round 6 (yh_conv_desc.bwd_z) inside a whole step on the GPU: YOLOv3 at 608 x 608, batch 3 (304^2 x 3 >= 262 144 pixels: the first
    residual stage's 1x1 data gradient runs on the persistent kernel and carries the backward sums of the block it completes), fp16, in
    the well-conditioned state (synth.calm_bn_: summation-order differences are not amplified).  Against the same step with
    YOLO_HIP_FUSE_DBN=0: the plan differs (a fused launch + a row-summing op in place of a reduction pass), every parameter gradient
    agrees to summation order - and twice in a row the fused step gives the same bits."""
    if DRY:
        pytest.skip('plan-level equivalence on the emulator: tests/test_train_emulated.py')
    import copy
    from models import Darknet
    from utils.utils import compute_loss
    import test_train_emulated as tte
    size, batch = 608, 3
    torch.manual_seed(0)
    model0 = Darknet(os.path.join(conftest.PKG, 'cfg', 'yolov3', 'yolov3.cfg'), (size, size))
    state = model0.state_dict()
    synth.calm_bn_(synth.randomize_bn_(state, seed=1))
    model0.load_state_dict(state)
    model0.nc, model0.hyp, model0.gr = 80, dict(tte.GOLD_HYP), 1.0
    targets = synth.loss_inputs(model0, size, batch=batch, seed=9, labels_per_image=8)[1].to(GPU)
    x = synth.image_batch(batch, size, seed=0).to(GPU)

    def step():
        model = copy.deepcopy(model0).train().to(GPU)
        monkeypatch.setenv('YOLO_HIP_TRAIN_PRECISION', 'fp16')
        pred, _ = model(x)
        monkeypatch.delenv('YOLO_HIP_TRAIN_PRECISION')
        eng = model.__dict__.get('_hip_train_engine')
        fused = [what for what, d in eng._current['bwd_ops'] if isinstance(d, hiplib.ConvDesc) and d.bwd_z]
        loss, _ = compute_loss(pred, targets, model)
        (loss * 1024.0).backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() / 1024.0 for k, p in model.named_parameters()}
        model.__dict__['_hip_train_engine'] = None
        return fused, grads
    fused, g1 = step()
    assert fused, 'no data gradient of this step carries backward sums'
    _, g1b = step()
    assert all(torch.equal(g1[k], g1b[k]) for k in g1), 'the fused step is not reproducible'
    monkeypatch.setenv('YOLO_HIP_FUSE_DBN', '0')
    plain, g0 = step()
    assert not plain
    num = sum(float((g1[k].double() - g0[k].double()).pow(2).sum()) for k in g1)
    den = sum(float(g0[k].double().pow(2).sum()) for k in g1)
    rel = (num / den) ** 0.5
    print('yolov3 608 b3 fp16: %d fused launches (%s); gradient of the fused step against the plain one: rel l2 %.2e' % (len(fused), ', '.join(fused), rel))
    assert rel <= 1e-3, rel


def test_fused_loss_defers_the_label_check_without_a_host_sync(libs):
    """The kernels record labels outside the class range on the device; the reference's AssertionError surfaces at the next
    compute_loss call / flush_label_check() instead of stalling the step on a device-to-host read."""
    if DRY:
        pytest.skip('the emulator raises immediately (tests/test_loss_golden.py)')
    from models import Darknet
    from utils import utils as U
    from engine import loss as hip_loss
    torch.manual_seed(0)
    model = Darknet(os.path.join(conftest.PKG, 'cfg', 'yolov3tiny/yolov3-tiny-hand.cfg'), (416, 416))
    model.nc, model.gr = 1, 1.0
    model.hyp = {'giou': 3.54, 'cls': 37.4, 'cls_pw': 1.0, 'obj': 64.3, 'obj_pw': 1.0, 'iou_t': 0.20, 'fl_gamma': 0.0}
    raws, targets = synth.loss_inputs(model, 416, batch=2, seed=3)
    raws = [r.to(GPU) for r in raws]
    bad = targets.clone()
    bad[:, 1] = 1.0
    hip_loss.flush_label_check()
    loss, _ = U.compute_loss(raws, bad.to(GPU), model)          # launches, does not raise
    with pytest.raises(AssertionError):
        hip_loss.flush_label_check()
    loss, items = U.compute_loss(raws, targets.to(GPU), model)  # a clean batch afterwards is unaffected
    hip_loss.flush_label_check()
    ref, ref_items = U.compute_loss([r.cpu() for r in raws], targets, model, fused=False)
    assert torch.allclose(items.cpu(), ref_items, rtol=1e-5, atol=1e-6)


def test_device_target_assignment_matches_build_targets_on_edge_cases(libs):
    """The loss kernels redo build_targets (reference utils.py:725-779) per candidate in fp32.  Labels sitting exactly on
    cell borders, boxes whose wh-IoU with an anchor is within one ulp of iou_t, many labels in one cell, tiny and
    image-sized boxes: loss items and gradients must equal the torch path on the CPU over many seeds."""
    if DRY:
        pytest.skip('the emulator reuses the torch assignment')
    from models import Darknet
    from utils import utils as U
    torch.manual_seed(0)
    model = Darknet(os.path.join(conftest.PKG, 'cfg', 'yolov3tiny/yolov3-tiny-hand.cfg'), (416, 416))
    model.nc, model.gr = 1, 0.5
    anchors = [model.module_list[j].anchor_vec.clone() for j in model.yolo_layers]
    grids = [416 // int(model.module_list[j].stride) for j in model.yolo_layers]
    for seed in range(12):
        model.hyp = {'giou': 3.54, 'cls': 37.4, 'cls_pw': 1.0, 'obj': 64.3, 'obj_pw': 1.0, 'iou_t': 0.20, 'fl_gamma': 0.0}
        g = torch.Generator().manual_seed(seed)
        raws, targets = synth.loss_inputs(model, 416, batch=3, seed=100 + seed, labels_per_image=10)
        t = targets.clone()
        n = t.shape[0]
        k = torch.randint(0, n, (8,), generator=g)
        cell = torch.randint(0, grids[seed % 2], (8, 2), generator=g).float()
        t[k, 2:4] = cell / grids[seed % 2]                       # centres exactly on cell corners
        t[k[:3], 2:4] = t[k[0], 2:4]                              # three labels in the same cell
        a = anchors[seed % 2][seed % 3] / grids[seed % 2]         # a box equal to an anchor (IoU exactly 1) ...
        t[k[3], 4:6] = a
        t[k[4], 4:6] = a * torch.tensor([0.2, 1.0])               # ... and one whose wh-IoU is iou_t up to rounding
        t[k[5], 4:6] = torch.tensor([1e-3, 1e-3])
        t[k[6], 4:6] = torch.tensor([0.999, 0.999])
        t[k[6], 2:4] = 0.5
        outs = []
        for dev, fused in ((GPU, True), (torch.device('cpu'), False)):
            ps = [r.clone().to(dev).requires_grad_() for r in raws]
            loss, items = U.compute_loss(ps, t.to(dev), model, fused=fused)
            loss.backward()
            outs.append((items.cpu(), [p.grad.cpu() for p in ps]))
        from engine import loss as hip_loss
        hip_loss.flush_label_check()
        assert torch.allclose(outs[0][0], outs[1][0], rtol=2e-5, atol=1e-6), (seed, outs[0][0], outs[1][0])
        for ga, gb in zip(outs[0][1], outs[1][1]):
            assert (ga - gb).abs().max().item() <= 1e-4 * gb.abs().max().item(), seed


# ------------------------------------------------------------------------------------------- widths that are not multiples of 8
@pytest.mark.parametrize('which', ['pruned_mini', 'odd', 'odd_mobile', 'ghost_like'])
@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
def test_odd_width_graphs_train_on_the_hip_path(libs, which, precision):
    """slim_prune-style graphs (arbitrary channel counts, incl. depthwise + squeeze-excite) train on the HIP kernels through the
    channel-padded twin (engine/padded.py) and match eager fp32 autograd on the unpadded modules; no eager fallback exists."""
    import models
    import test_plan_emulated as tpe
    import test_train_emulated as tte
    from engine.padded import PaddedTrainEngine
    if which in ('pruned_mini', 'ghost_like'):   # ghost_like: a 24-channel conv added to a 12 + 12 concat (GhostNet's join)
        path = th.write_cfg(tte._pruned_like_cfg_text() if which == 'pruned_mini' else tte._ghost_like_text())
        model = th.build(path, 64)
        os.unlink(path)
    else:
        defs = {'odd': tpe._odd_cfg, 'odd_mobile': tpe._odd_mobile_cfg}[which]()
        torch.manual_seed(5)
        model = models.Darknet(defs, (64, 64))
        model.load_state_dict(synth.randomize_bn_(model.state_dict(), seed=6))
        model.train()
    x = synth.image_batch(3, 64, seed=0 if which == 'pruned_mini' else 7)   # samples on which no activation sits on a kink
    raws_ref, grads_ref, m_ref, ws = th.eager_step(model, x)
    raws, grads, m = th.engine_step(model, x, ws, precision, lib=fakelib.FakeLib() if DRY else None, device=GPU)
    assert isinstance(m.__dict__['_hip_train_engine'], PaddedTrainEngine)
    total = sum(g.norm().item() ** 2 for g in grads_ref.values()) ** 0.5
    if precision == 'fp32':
        for a, b in zip(raws, raws_ref):
            assert (a - b).abs().max().item() <= 5e-5 * b.abs().max().item()
        for k in grads_ref:
            # random-weight leaky / relu6 nets: a kink that flips under another fp32 summation order moves upstream gradients by 1e-3
            assert (grads[k] - grads_ref[k]).norm().item() <= 5e-3 * grads_ref[k].norm().item() + 1e-5 * total, k
        num = sum((grads[k] - grads_ref[k]).norm().item() ** 2 for k in grads_ref) ** 0.5
        assert num <= 2e-3 * total, (num, total)
        for (k, a), (_, b) in zip(m.state_dict().items(), m_ref.state_dict().items()):
            if 'running' in k:
                assert (a.cpu() - b).abs().max().item() <= 1e-5 * (b.abs().max().item() + 1), k
    else:
        num = sum((grads[k] - grads_ref[k]).norm().item() ** 2 for k in grads_ref) ** 0.5
        dot = sum((grads[k].double() * grads_ref[k].double()).sum().item() for k in grads_ref)
        mine = sum(g.double().norm().item() ** 2 for g in grads.values()) ** 0.5
        # fp16 storage of every activation / gradient on a tiny 13 - 30 channel net: direction and size of the whole gradient
        assert dot / (mine * total) >= 0.99 and num <= 0.15 * total, (num, total, dot / (mine * total))


def test_training_forward_has_no_eager_fallback(libs):
    """A GPU training forward either runs on the HIP step or raises: feature_out (the feature-distillation losses, out of
    scope) is refused instead of switching to the eager modules; a cfg the step cannot lower raises too (GhostNet-style gathered
    shortcuts between differently padded tensors)."""
    if DRY:
        pytest.skip('needs a GPU')
    import models
    model = models.Darknet(os.path.join(conftest.PKG, 'cfg', 'yolov3tiny', 'yolov3-tiny-hand.cfg'), (64, 64)).cuda().train()
    x = synth.image_batch(2, 64, seed=1).cuda()
    model.hip_return_features = True
    with pytest.raises(NotImplementedError, match='feature_out'):
        model(x)
    model.hip_return_features = False
    out, feats = model(x)
    assert feats == [] and model.__dict__['_hip_train_engine'] is not None


# ------------------------------------------------------------------------- BASELINE config 5 at its real shape
def test_reference_pruned_mobilenet_fine_tunes_at_416():
    """The compact graph the reference's UNMODIFIED slim_prune.py --percent 0.5 wrote for yolov3-mobilenet-coco (tools/make_pruned.py;
    cfg text in tests/golden/) - 70 conv blocks, 37 widths off the 8-channel grid, depthwise + squeeze-excite - trains at 416 px on
    the HIP path through the channel-padded twin, against eager fp64 autograd (reference slim_prune.py:97-207 -> train.py fine-tune).

    * Mish in place of the kinked activations (same graph, same weights): the lowering must agree with fp64 as closely as eager
      fp32 autograd does (round-off).
    * As pruned (relu6 / h-swish / leaky): a pre-activation within rounding distance of a kink lands on either side depending on
      the summation order; the sample's own kink sensitivity - the change of the fp64 gradient when the frames move by one fp32
      ulp - bounds what ANY fp32 implementation can promise.  Round 2 saw 8.55e-3 on one sample where eager fp32 had 1.1e-5:
      that sample's sensitivity is of the same size (profiles/r03_pruned_mobilenet_finetune.txt), i.e. eager was lucky, the
      twin is not defective."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    'yolov3v4-modelcompression-multidatasettraining-multibackbone_amd', 'tools'))
    import pruned_finetune as pf
    from engine.padded import PaddedTrainEngine
    size = 416
    model = pf.build(pf.GOLD_CFG, '', size)
    widths = [b[0].out_channels for b in model.module_list if isinstance(b, torch.nn.Sequential) and hasattr(b[0], 'out_channels')]
    assert len(widths) == 70 and sum(1 for c in widths if c % 8) >= 30
    smooth = pf.smooth_twin(model, pf.GOLD_CFG, size)
    x = torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(4))
    e32, ehip, sens, m = pf.sample_report(smooth, x, 'cuda')
    assert isinstance(m.__dict__['_hip_train_engine'], PaddedTrainEngine)
    print('pruned mobilenet 416, Mish twin: eager fp32 %.2e, HIP %.2e, sensitivity %.2e' % (e32, ehip, sens))
    assert ehip <= 4 * e32 + 2e-5, (e32, ehip)
    for seed in (4, 6):
        x = torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(seed))
        e32, ehip, sens, _ = pf.sample_report(model, x, 'cuda')
        print('pruned mobilenet 416, as pruned, sample %d: eager fp32 %.2e, HIP %.2e, fp64 kink sensitivity %.2e' % (seed, e32, ehip, sens))
        assert ehip <= 4 * max(e32, sens) + 2e-5, (seed, e32, ehip, sens)
