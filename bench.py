#!/usr/bin/env python3
"""Benchmark of the Darknet hot path on MI355X (BASELINE.json: "images/sec YOLOv3-608 train fp16 @1/2/4/8 GPU; detect
fp16/int8 FPS").

    python bench.py --gpus N --steps K --warmup W [--mode both|train|detect] [--precision fp16|fp32|int8]

Default (--mode both): the headline `value` is the TRAINING step (configs[2] shape: YOLOv3-608, batch 64 per GPU, fp16
autocast with fp32 master weights) and the detection leg (configs[1]: forward + NMS) rides along under "detect".

* train step = train-mode forward (batch-statistics BatchNorm) + compute_loss on synthetic labels + backward + GradScaler
  + nesterov SGD (reference train.py:371-433), all convolution / BatchNorm / activation work on libyolo_hip.so.
  N > 1: one process per GPU, DistributedDataParallel over RCCL, per-GPU batch fixed (weak scaling), gradients
  all-reduced by DDP's hooks.
* detect step = detect.py's loop body on a batch already resident in HBM: Darknet eval forward (BN folded, fused epilogues),
  3-scale YOLO decode and NMS at detect.py's settings (conf 0.3, iou 0.6, best class) - since round 5 as model.hip_detect (decode
  fused into the NMS candidate filter, same detections bit for bit; --two-pass-nms: model(x) then non_max_suppression, rounds 1 - 4).
  N > 1: replicas only (no exchange).

Synthetic data and random-init weights (no network access).  Timing: W warm-up steps, then exactly K steps bracketed
by barrier + synchronize on both sides, MAX over ranks (engine/distutil.py).  Rank 0 prints ONE JSON line; besides the
driver contract it carries
  roofline      the dominant kernel class of the measured step (per-op HIP events recorded by the native plan executor
                on the launch stream): algorithmic conv FLOPs / summed duration against the 2.5 PFLOP/s dense fp16 peak,
                plus the per-kernel-class table of the step;
  roofline_net  whole-step conv FLOPs/s from the un-instrumented timed region (forward + dgrad + wgrad = 3 x 140.7 GFLOP
                per image for training);
  cpu_baseline  the same step on this host's cores (eager fp32 torch modules, bit-equal to the reference's) on a bounded
                sample (N = 1 only);
  detect        metric/value/roofline of the fp16 detection leg;
  detect_int8   the same for the int8 COS-PTQ graph (synthetic power-of-two calibration state, tools/synthetic_ptq.py).
If the training leg cannot run, the detection leg becomes the headline and "train_error" says why.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, 'yolov3v4-modelcompression-multidatasettraining-multibackbone_amd')
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # the host driver only supports dmabuf IPC (RCCL needs it)

import torch  # noqa: E402

PEAK_TFLOPS = {'fp16': 2500.0, 'fp32': 157.3, 'int8': 5000.0}  # dense MFMA peaks (int8: TOP/s), MI355X_MICROARCH.md
TILE_NAMES = {1: '128x128', 2: '64x256', 3: '32x256', 4: '64x128', 5: '128x64', 6: '256x128', 11: '128x128k8', 12: '64x256k8',
              14: '64x128k8', 15: '128x64k8', 16: '256x128k8', 21: 'dma3_128x128', 22: 'dma3_64x256', 24: 'dma3_64x128',
              25: 'dma3_128x64', 26: 'dma3_256x128', 27: 'dma3_128x256', 41: 'halo_128x256', 43: 'hpp_128x512', 51: 'abl_noload', 52: 'abl_nomfma', 42: 'halo_256x256', 31: 'dma4_128x128', 32: 'dma4_64x256', 34: 'dma4_64x128',
              35: 'dma4_128x64', 61: 'k64_256x256', 62: 'k64_128x512', 63: 'k64_256x128w4', 64: 'pp_256x256', 65: 'pp_128x512',
              66: 'pp_512x128', 67: 'pp2_256x256', 68: 'pp2_128x512', 69: 'pp2_512x128', 71: 'pw_stream', 72: 's3_stream', 73: 'pwl_stream'}


def tile_name(code):
    return TILE_NAMES.get(code, 'tile%d' % code)


def conv_tile(lib, desc):
    """Tile code the launch of a RECORDED plan op will use: the statistics workspace of a training-forward conv is bound per run
    (a plan slot), so the recorded descriptor has stats_ws == NULL with stats_ws_floats > 0 - and kernels without a statistics
    epilogue must not be named for it."""
    import ctypes as C
    if getattr(desc, 'stats_ws_floats', 0) > 0 and not desc.stats_ws:
        twin = type(desc)()
        C.memmove(C.byref(twin), C.byref(desc), C.sizeof(desc))
        twin.stats_ws = 4          # any non-null address: the query only tests it
        desc = twin
    return lib.yh_conv2d_tile(C.byref(desc))


def host_threads(budget_s=4.0):
    """Thread count for the CPU baseline: BASELINE.md 3 asks for every host core, but on the 256-CPU GPU node a training step on all
    256 threads took 222 s (oneDNN / OpenMP oversubscription behind the container's CPU quota) against 13 s on 64.  So the
    candidates {all usable CPUs, 1/2, 1/4, 64, 32} are probed with one conv layer each and the FASTEST is used - the best this
    host can do - with the probe table reported next to the number."""
    import torch.nn.functional as F
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    try:   # cgroup v2 CPU quota
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    cands = sorted({c for c in (usable, usable // 2, usable // 4, 64, 32) if 1 <= c <= usable}, reverse=True)
    x = torch.rand(2, 64, 152, 152)
    w = torch.rand(128, 64, 3, 3)
    table = {}
    for c in cands:
        torch.set_num_threads(c)
        F.conv2d(x, w, padding=1)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s / len(cands):
            F.conv2d(x, w, padding=1)
            n += 1
        table[c] = round(n / (time.perf_counter() - t0), 1)
    best = max(table, key=lambda c: table[c])
    torch.set_num_threads(best)
    return best, usable, table


def model_label(cfg, size):
    """'YOLOv3-608' for cfg/yolov3/yolov3.cfg at 608: the label follows --cfg / --size, whatever they are."""
    base = os.path.splitext(os.path.basename(cfg))[0]
    pretty = {'yolov3': 'YOLOv3', 'yolov4': 'YOLOv4', 'yolov3-tiny': 'YOLOv3-tiny', 'yolov4-tiny': 'YOLOv4-tiny',
              'yolov3-mobilenet-coco': 'YOLOv3-Mobilenetv3', 'yolov3-spp': 'YOLOv3-SPP'}.get(base, base)
    return '%s-%d' % (pretty, size)


def model_family(cfg):
    base = os.path.splitext(os.path.basename(cfg))[0]
    return {'yolov3': 'YOLOv3 Darknet-53', 'yolov4': 'YOLOv4 CSPDarknet53 + Mish + SPP/PAN'}.get(base, base)


def build_model(cfg, size, precision, device):
    import models
    torch.manual_seed(0)
    model = models.Darknet(cfg, (size, size))
    g = torch.Generator().manual_seed(1)
    state = model.state_dict()
    for k, v in state.items():
        if k.endswith('running_var') or k.endswith('BatchNorm2d.weight'):
            v.copy_(torch.rand(v.shape, generator=g) + 0.5)
        elif k.endswith('running_mean') or k.endswith('BatchNorm2d.bias'):
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    model.load_state_dict(state)
    model.hip_precision = precision
    return model.to(device).eval()


def build_qmodel_synthetic(cfg, size, device, float_model=None, frames=None):
    """A COS-PTQ graph (Darknet(quantized=3)) with a synthetic calibrated state, for int8 *timing* only.

    Weight/bias grids come from the BN-folded seeded float weights with power-of-two max-abs scales; every
    activation, shortcut and concat scale is the power of two that just covers the float model's range of that block on
    `frames` (what a one-batch max-abs calibration would see: a live int8 detector whose heads follow the float model's, so
    that NMS gets the same kind of candidate set), or a fixed power of two when no frames are given.  The module classes are this package's
    utils/quantized/quantized_ptq_cos.py (same names, buffers and eval arithmetic as the reference's; pinned to it in
    tests/test_ptq.py and tests/test_ptq_calibration.py).  Only the HIP int8 engine is timed."""
    import models
    from tools.synthetic_ptq import fill_synthetic_state
    fm = float_model.cpu() if float_model is not None else build_model(cfg, size, 'fp16', 'cpu')
    torch.manual_seed(0)
    qm = models.Darknet(cfg, (size, size), quantized=3, a_bit=8, w_bit=8, shortcut_way=1)
    ranges = None
    if frames is not None:
        from tools.synthetic_ptq import measure_ranges
        ranges = measure_ranges(fm, frames.detach().float().cpu())
    fill_synthetic_state(fm, qm, ranges=ranges)
    return qm.to(device).eval()


def build_qmodel_calibrated(cfg, size, device, float_model, frames, batches=3, batch=2):
    """The COS-PTQ graph calibrated ON THE DEVICE as PTQ.py does it (reference PTQ.py:60-100: train-mode forwards over calibration
    batches; every scale vote through csrc/calib.hip, every calibration convolution through yh_conv2d_fwd fp32): `batches` batches of
    `batch` frames of the bench input.  tests/test_ptq_calibration.py::test_device_calibration_at_the_baseline_shapes_then_int8_engine
    pins this flow (every search equals the reference's loop on the same tensor; the calibrated int8 engine equals the calibrated
    modules' CPU evaluation)."""
    import models
    fm = float_model
    torch.manual_seed(0)
    qm = models.Darknet(cfg, (size, size), quantized=3, a_bit=8, w_bit=8, shortcut_way=1)
    with torch.no_grad():     # load_darknet_weights(quant=True) semantics (reference models.py:610-628): BN tensors land on the conv
        for f, q in zip(fm.module_list, qm.module_list):
            if isinstance(f, torch.nn.Sequential) and len(f) and isinstance(f[0], torch.nn.Conv2d):
                qc = q[0]
                qc.weight.data.copy_(f[0].weight.data.float())
                bn = f[1] if len(f) > 1 and isinstance(f[1], torch.nn.modules.batchnorm.BatchNorm2d) else None
                if bn is not None:
                    qc.gamma.data.copy_(bn.weight.data.float())
                    qc.beta.data.copy_(bn.bias.data.float())
                    qc.running_mean.copy_(bn.running_mean.float())
                    qc.running_var.copy_(bn.running_var.float())
                else:
                    qc.bias.data.copy_(f[0].bias.data.float())
                    qc.gamma.data.zero_()
                    qc.beta.data.zero_()
    qm.to(device).train()
    with torch.no_grad():
        for it in range(batches):
            lo = (it * batch) % max(1, frames.shape[0] - batch + 1)
            qm(frames[lo:lo + batch].float())
    return qm.eval()


def conv_flops(plan):
    """Algorithmic FLOPs (2 x MACs on logical channels) of every conv op of a built plan, by op index."""
    flops = {}
    values = {('conv%d' % v.block if v.src.kind != 'input' else 'stem%d' % v.block): v
              for v in plan['values'] if v.kind == 'conv'}
    for idx, (what, desc) in enumerate(plan['ops']):
        v = values.get(what)
        if v is not None:
            flops[idx] = 2.0 * plan['N'] * v.Ho * v.Wo * v.C * v.k * v.k * v.src.C
    return flops


def roofline_leg(model, x, steps, precision):
    from engine import hiplib
    eng = model.__dict__['_hip_engine']
    plan = eng._plans[tuple(x.shape)]
    lib, handle = eng.lib, plan['handle']
    n_ops = lib.yh_plan_num_ops(handle)
    hiplib.check(lib.yh_plan_set_timing(handle, 1), 'set_timing')
    buf = (C.c_float * n_ops)()
    total = [0.0] * n_ops
    t0 = time.perf_counter()
    for _ in range(steps):
        with torch.no_grad():
            model(x)
        torch.cuda.synchronize()
        hiplib.check(lib.yh_plan_get_timings(handle, buf, n_ops), 'get_timings')
        for i in range(n_ops):
            total[i] += buf[i]
    wall = (time.perf_counter() - t0) / steps
    hiplib.check(lib.yh_plan_set_timing(handle, 0), 'set_timing')
    flops = conv_flops(plan)
    groups = {}
    for idx, (what, desc) in enumerate(plan['ops']):
        ms = total[idx] / steps
        if isinstance(desc, hiplib.ConvDesc):
            name = 'conv_igemm_%s_%s' % (precision, tile_name(conv_tile(lib, desc)))
        else:
            name = ''.join(c for c in what if not c.isdigit())
        g = groups.setdefault(name, dict(ms=0.0, flops=0.0, launches=0, bytes=[]))
        g['ms'] += ms
        g['flops'] += flops.get(idx, 0.0)
        g['launches'] += 1
        if isinstance(desc, hiplib.ConvDesc):
            g['bytes'].append(op_bytes('conv', desc, 1 if precision == 'int8' else (2 if precision == 'fp16' else 4)))
    dom_name = max((n for n in groups if n.startswith('conv_igemm')), key=lambda n: groups[n]['ms'])
    dom = groups[dom_name]
    achieved = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
    peak = PEAK_TFLOPS[precision]
    breakdown = {n: dict(ms=round(g['ms'], 4), launches=g['launches'],
                         tflops=round(g['flops'] / (g['ms'] * 1e-3) / 1e12, 1) if g['flops'] else None)
                 for n, g in sorted(groups.items(), key=lambda kv: -kv[1]['ms'])}
    return dict(bound='mfma', kernel=dom_name, achieved=round(achieved, 2), peak=peak, unit='TFLOP/s',
                frac=round(achieved / peak, 4), traffic=hbm_traffic(dom_name, x.shape[0], dom['bytes']), launches_per_step=dom['launches'],
                avg_launch_ms=round(dom['ms'] / dom['launches'], 5),
                gflop_per_launch=round(dom['flops'] / dom['launches'] / 1e9, 3),
                gpu_ms_per_step_all_kernels=round(sum(g['ms'] for g in groups.values()), 4),
                instrumented_ms_per_step=round(wall * 1e3, 4), kernels=breakdown)


def hbm_traffic(kernel, batch, alg_bytes=None):
    """HBM bytes per launch of the dominant detect kernel from the committed rocprofv3 PMC passes (see traffic_of), or None."""
    key = rocprof_kernel_name(kernel)
    section = 'batch%d%s' % (batch, '_int8' if '_int8_' in kernel else '')
    return traffic_of(key, section, alg_bytes) if key else None


def rocprof_kernel_name(kernel):
    """bench.py's kernel label (conv_igemm_<precision>_<tile name>) -> the name tools/rocprof_summary.py prints for that instantiation."""
    import re
    m = re.match(r'conv_igemm_(fp16|int8|fp32)_(dma3|halo|pp|hpp)_(\d+x\d+)$', kernel)
    if not m:
        m2 = re.match(r'conv_igemm_(fp16|int8)_(pwl_stream|s3_stream|pw_stream)$', kernel)
        if m2:
            t = {'fp16': 'f16', 'int8': 'i8'}[m2.group(1)]
            return {'pwl_stream': 'conv1x1_lds_kernel', 's3_stream': 'conv3x3_stream<%s>' % t, 'pw_stream': 'conv_pointwise_kernel'}[m2.group(2)]
        return None
    t = {'fp16': 'f16', 'int8': 'i8', 'fp32': 'f32'}[m.group(1)]
    return {'dma3': 'conv_igemm_glds<%s,%s,%s,S3>', 'halo': 'conv3x3_halo<%s,%s,%s>', 'pp': 'conv_igemm_pp<%s,%s,%s>',
            'hpp': 'conv3x3_hpp<%s,%s,%s>'}[m.group(2)] % (t, t, m.group(3))


def cpu_baseline(cfg, size, budget_s):
    """The oracle (CPU port of the reference path) on this host: YOLOv3-608, batch 1, fp32, BN folded."""
    import models
    from oracle import darknet_oracle as oracle
    torch.manual_seed(0)
    m = models.Darknet(cfg, (size, size)).eval()
    state = m.state_dict()
    x = torch.rand(1, 3, size, size)
    threads, usable, probe = host_threads()
    with torch.no_grad():
        oracle.forward(m.module_defs, state, x)  # warm-up (page-in, thread pool)
        n, t0 = 0, time.perf_counter()
        while True:
            oracle.forward(m.module_defs, state, x)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= budget_s or n >= 64:
                break
    return dict(value=round(n / dt, 3), unit='images/s', cores=torch.get_num_threads(), kind='port',
                sample='%d images of %s, batch 1, fp32, oracle.forward (BN folded), %.1f s on %d threads (host: %d CPUs, %d usable; '
                       'conv probe it/s by thread count %s)' % (n, model_label(cfg, size), dt, threads, os.cpu_count() or 1, usable, probe))


HYP = {'giou': 3.54, 'cls': 37.4, 'cls_pw': 1.0, 'obj': 64.3, 'obj_pw': 1.0, 'iou_t': 0.20, 'lr0': 0.01, 'momentum': 0.937,
       'weight_decay': 0.0005, 'fl_gamma': 0.0}  # reference train.py:25-35


def synthetic_labels(batch, per_image, nc, seed, device):
    g = torch.Generator().manual_seed(seed)
    n = batch * per_image
    wh = torch.rand(n, 2, generator=g) * 0.4 + 0.03
    xy = torch.rand(n, 2, generator=g) * (1 - wh) + wh / 2
    img = torch.arange(batch).repeat_interleave(per_image).float().view(-1, 1)
    cls = torch.randint(0, nc, (n, 1), generator=g).float()
    return torch.cat((img, cls, xy, wh), 1).to(device)


def train_main(args, device, dist, world, rank, local_rank):
    """configs[2]: one optimisation step = train-mode forward + compute_loss + backward + SGD (reference train.py:371-433:
    autocast forward, loss scaled by batch/64, GradScaler backward, nesterov SGD with the three parameter groups)."""
    from models import Darknet
    from utils.utils import compute_loss
    from engine import distutil
    if args.precision == 'int8':
        raise SystemExit('--mode train runs fp16 (autocast, fp32 master weights) or fp32')
    torch.manual_seed(0)
    model = Darknet(args.cfg, (args.size, args.size)).to(device)
    pg0, pg1, pg2 = [], [], []
    for k, v in dict(model.named_parameters()).items():   # train.py:112-119
        if '.bias' in k:
            pg2.append(v)
        elif 'Conv2d.weight' in k:
            pg1.append(v)
        else:
            pg0.append(v)
    # same update rule as train.py:121; torch's single-launch multi-tensor implementation when the build has it
    sgd_impl = os.environ.get('YOLO_BENCH_SGD', 'fused')
    try:
        opt = torch.optim.SGD(pg0, lr=HYP['lr0'] * 0.01, momentum=HYP['momentum'], nesterov=True, fused=(sgd_impl == 'fused'))
    except (RuntimeError, TypeError):
        opt = torch.optim.SGD(pg0, lr=HYP['lr0'] * 0.01, momentum=HYP['momentum'], nesterov=True)
    opt.add_param_group({'params': pg1, 'weight_decay': HYP['weight_decay']})
    opt.add_param_group({'params': pg2})
    core = model
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
                                                          bucket_cap_mb=args.bucket_mb, gradient_as_bucket_view=True)
        if args.grad_compress == 'fp16':
            # halve the xGMI payload (248 MB -> 124 MB per step for YOLOv3): buckets are cast to fp16 for the all-reduce and back
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
            model.register_comm_hook(None, default_hooks.fp16_compress_hook)
        model.yolo_layers = core.yolo_layers
    for m in {model, core}:
        m.nc, m.hyp, m.gr = 80, HYP, 1.0
    model.train()
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.rand(args.batch, 3, args.size, args.size, generator=g).to(device)
    targets = synthetic_labels(args.batch, 8, 80, 200 + rank, device)
    fp16 = args.precision == 'fp16'
    scaler = torch.amp.GradScaler('cuda', enabled=fp16)
    state = {}

    def step():
        with torch.autocast('cuda', dtype=torch.float16, enabled=fp16):
            pred, _ = model(x)
        loss, items = compute_loss(pred, targets, model)
        loss = loss * (args.batch / 64)
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        state['loss'] = items

    for _ in range(args.warmup):
        step()
    eng = core.__dict__.get('_hip_train_engine')
    if eng is None:
        raise SystemExit('the HIP training path did not engage (cfg not lowered?)')
    elapsed = distutil.timed_region(step, args.steps, dist, device)
    if rank == 0:
        images = world * args.batch * args.steps
        value = images / elapsed
        plan = eng._current
        fwd_flops = sum(2.0 * plan['N'] * v.Ho * v.Wo * v.C * v.conv.in_channels * v.k * v.k
                        for v in plan['values'] if v.kind == 'conv')
        gflop_img = 3 * fwd_flops / args.batch / 1e9      # forward + data gradient + weight gradient
        peak = PEAK_TFLOPS[args.precision]
        net = value / world * gflop_img / 1e3
        out = {
            'metric': 'images/sec %s train %s' % (model_label(args.cfg, args.size), args.precision), 'value': round(value, 2), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'fp16': 'f16', 'fp32': 'f32'}[args.precision], 'data': 'synthetic',
            'config': {'workload': '%s %d COCO (80 classes) %s training step, batch %d/GPU: train-mode forward '
                                   '(batch-stat BN) + compute_loss + backward + nesterov SGD, %s'
                                   % (model_family(args.cfg), args.size, args.precision, args.batch,
                                      ('DDP gradient all-reduce over %s' % ('RCCL' if args.dist_backend == 'nccl' else 'gloo (host memory)'))
                                      if world > 1 else 'single GPU'),
                       'global_batch': world * args.batch, 'parallelism': 'dp%d' % world, 'gflop_per_image': round(gflop_img, 2),
                       'loss': [round(float(v), 4) for v in state['loss']]},
            'roofline_net': {'bound': 'mfma', 'achieved': round(net, 2), 'peak': peak, 'unit': 'TFLOP/s',
                             'frac': round(net / peak, 4), 'per': 'GPU, whole step incl. loss, optimizer and host gaps'},
        }
        out['roofline'] = train_roofline(eng, x, args.precision)
        out['cpu_baseline'] = None if (world > 1 or args.no_cpu_baseline) else cpu_train_baseline(args.cfg, args.cpu_seconds, args.size)
    else:
        out = None
    distutil.barrier(dist)
    core.__dict__['_hip_train_engine'] = None   # release the ~40 GB of step buffers before the next leg
    del model, core, opt, eng
    torch.cuda.empty_cache()
    return out


WGRAD_KERNELS = {90: 'conv_wgrad_halo', 91: 'conv_wgrad_roll', 22: 'conv_wgrad_dma<2,2>', 42: 'conv_wgrad_dma<4,2>', 44: 'conv_wgrad_dma<4,4>',
                 82: 'conv_wgrad_dma<8,2>', 84: 'conv_wgrad_dma<8,4>', 1: 'conv_wgrad'}   # yh_conv2d_wgrad_kernel codes


def op_bytes(role, desc, esz):
    """Algorithmic HBM bytes of one conv-type launch: every operand once (activations in, residual, out, weights / fp32 dW)."""
    if role == 'wgrad':
        return (float(desc.n) * desc.h * desc.w_in * desc.cin + float(desc.n) * desc.ho * desc.wo * desc.cout) * esz \
            + 4.0 * desc.cout * (desc.cin_w or desc.cin) * desc.kh * desc.kw
    out_px = float(desc.n) * desc.ho * desc.wo * (4 if desc.ups == 2 else 1)
    out_c = desc.cout // 4 if desc.ups == 4 else desc.cout
    if desc.ups == 4:
        out_px *= 4
    b = float(desc.n) * desc.h * desc.w_in * desc.cin * esz + out_px * out_c * (4 if desc.out_f32 else esz)
    if desc.res:
        b += out_px * out_c * esz
    return b + float(desc.cout) * desc.cin * desc.kh * desc.kw * esz


def traffic_of(kernel, section, launches_alg_bytes=None):
    """HBM counters of one kernel instantiation from the committed rocprofv3 PMC passes (profiles/hbm_traffic.json, written by
    tools/rocprof_summary.py traffic from separate --pmc FETCH_SIZE / WRITE_SIZE runs of this command), with the algorithmic
    bytes of the same launches next to it.  WRITE_SIZE is exact on this part (calibrated, profiles/r04_traffic_calibration.txt);
    FETCH_SIZE counts a 128-byte request as 64 bytes and a 64-byte request as 64, so the truth lies between `fetch_raw` and
    2 x `fetch_raw`: `hbm_bytes_per_launch` uses the guide's doubling (an upper bound for kernels that fetch 64-byte row pieces)."""
    try:
        table = json.load(open(os.path.join(REPO, 'profiles', 'hbm_traffic.json')))
    except Exception:
        return None
    ent = table.get(section, {}).get(kernel) if kernel else None
    if ent is None:
        return None
    out = {'kernel': kernel, 'hbm_bytes_per_launch': ent['hbm_bytes_per_dispatch'], 'write_bytes': int(ent['write_kib'] * 1024),
           'fetch_bytes_raw': int(ent['fetch_kib_raw'] * 1024), 'launches_profiled': ent.get('dispatches'),
           'source': 'profiles/hbm_traffic.json[%s]' % section}
    if launches_alg_bytes:
        alg = sum(launches_alg_bytes) / len(launches_alg_bytes)
        out['algorithmic_bytes_per_launch'] = int(alg)
        out['ratio_to_algorithmic'] = [round((out['write_bytes'] + out['fetch_bytes_raw']) / alg, 2), round(out['hbm_bytes_per_launch'] / alg, 2)]
    return out


def train_roofline(eng, x, precision):
    """Per-op HIP-event timing of one forward + backward plan replay.  Kernels are grouped by INSTANTIATION (`by_kernel`) and by
    class (all weight-gradient kernels form one class; every implicit-GEMM tile is its own); `roofline` is quoted on the class
    that takes the most time, names the instantiation that dominates inside it and attaches THAT kernel's counters."""
    import ctypes as C
    lib, plan = eng.lib, eng._current
    heads = eng.forward(x)
    torch.cuda.synchronize()
    from engine import hiplib
    esz = 2 if precision == 'fp16' else 4
    groups, roles = {}, {}
    for key, log in (('fwd', plan['fwd_ops']), ('bwd', plan['bwd_ops'])):
        handle = plan[key]
        lib.yh_plan_set_timing(handle, 1)
        if key == 'fwd':
            eng.forward(x)
        else:
            eng.backward([torch.randn_like(h) * 1e-3 for h in heads])
        torch.cuda.synchronize()
        n = lib.yh_plan_num_ops(handle)
        buf = (C.c_float * n)()
        lib.yh_plan_get_timings(handle, buf, n)
        lib.yh_plan_set_timing(handle, 0)
        for (what, desc), ms in zip(log, buf):
            role = what.rstrip('0123456789')
            # group by KERNEL instantiation: the forward convs and the data gradients share the implicit-GEMM kernels
            if isinstance(desc, hiplib.ConvDesc):
                name = 'conv_igemm_%s_%s' % (precision, tile_name(conv_tile(lib, desc)))
                cls = name
            elif role == 'wgrad':
                code = int(lib.yh_conv2d_wgrad_kernel(C.byref(desc)))
                name, cls = WGRAD_KERNELS.get(code, 'conv_wgrad_%d' % code), 'conv_wgrad'
            else:
                name = cls = role
            grp = groups.setdefault(name, dict(ms=0.0, flops=0.0, n=0, cls=cls, bytes=[]))
            grp['ms'] += ms
            grp['n'] += 1
            if role in ('conv', 'dgrad', 'wgrad'):
                grp['flops'] += 2.0 * desc.n * desc.ho * desc.wo * desc.cout * desc.cin * desc.kh * desc.kw
                grp['bytes'].append(op_bytes(role, desc, esz))
            roles.setdefault(role, [0.0])[0] += ms
    total = sum(g['ms'] for g in groups.values())
    section = 'train_batch%d' % x.shape[0]

    def rocprof_name(k):
        return k if k.startswith('conv_wgrad') else rocprof_kernel_name(k)

    table = {}
    for k, g in sorted(groups.items(), key=lambda kv: -kv[1]['ms']):
        table[k] = dict(ms=round(g['ms'], 3), n=g['n'], tflops=round(g['flops'] / g['ms'] / 1e9, 1) if g['ms'] > 0 else 0)
        if g['bytes']:
            table[k]['algorithmic_mb_per_launch'] = round(sum(g['bytes']) / len(g['bytes']) / 1e6, 1)
            table[k]['algorithmic_tb_s'] = round(sum(g['bytes']) / g['ms'] / 1e9, 2)
    classes = {}
    for k, g in groups.items():
        c = classes.setdefault(g['cls'], dict(ms=0.0, flops=0.0, n=0, members=[]))
        c['ms'] += g['ms']
        c['flops'] += g['flops']
        c['n'] += g['n']
        c['members'].append(k)
    top = max((c for c in classes if classes[c]['flops'] > 0), key=lambda c: classes[c]['ms'])
    cg = classes[top]
    dom = max(cg['members'], key=lambda k: groups[k]['ms'])          # the instantiation that dominates the class by time
    ach = cg['flops'] / cg['ms'] / 1e9
    peak = PEAK_TFLOPS[precision]
    dg = groups[dom]
    traffic = traffic_of(rocprof_name(dom), section, dg['bytes']) if precision == 'fp16' else None
    if traffic is not None and dom in ('conv_wgrad_halo', 'conv_wgrad_roll'):
        companion = traffic_of('wgrad_halo_reduce' if dom == 'conv_wgrad_halo' else 'wgrad_roll_reduce', section)
        if companion:
            traffic['second_launch'] = {k: companion[k] for k in ('kernel', 'hbm_bytes_per_launch', 'write_bytes', 'fetch_bytes_raw')}
    members = {k: dict(ms=round(groups[k]['ms'], 3), n=groups[k]['n'], tflops=round(groups[k]['flops'] / groups[k]['ms'] / 1e9, 1),
                       frac=round(groups[k]['flops'] / groups[k]['ms'] / 1e9 / peak, 4)) for k in sorted(cg['members'], key=lambda k: -groups[k]['ms'])}
    by_class = {c: dict(ms=round(v['ms'], 3), n=v['n'], tflops=round(v['flops'] / v['ms'] / 1e9, 1), frac=round(v['flops'] / v['ms'] / 1e9 / peak, 4))
                for c, v in sorted(classes.items(), key=lambda kv: -kv[1]['ms']) if v['flops'] > 0}
    # flat scalars first: the driver's record keeps the scalar keys of this object and drops the nested ones - a
    # change of the quoted class between rounds, and the share of the BatchNorm passes, stay visible without profiles/
    ranked = list(by_class.items())
    second = ranked[1] if len(ranked) > 1 else (None, {})
    bn_ms = round(sum(v[0] for k, v in roles.items() if k in ('bnact', 'dbn', 'dbnx', 'bnfin')), 3)
    return {'bound': 'mfma', 'kernel': dom, 'kernel_class': top,
            'achieved': round(ach, 1), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
            'traffic': None if traffic is None else traffic['hbm_bytes_per_launch'],
            'class_ms': round(cg['ms'], 3), 'class2': second[0], 'class2_ms': second[1].get('ms'), 'class2_tflops': second[1].get('tflops'),
            'class2_frac': second[1].get('frac'), 'batchnorm_passes_ms': bn_ms,
            'top_classes': '; '.join('%s %.2f ms %.0f TFLOP/s %.3f' % (c, v['ms'], v['tflops'], v['frac']) for c, v in ranked[:4]),
            'class_members': members, 'by_class': by_class,
            'quoted_on': 'the kernel class with the largest share of the step (all its launches); `kernel` is the instantiation that '
                         'dominates that class by time, `traffic` are that kernel\'s own counters',
            'traffic_detail': traffic, 'launches_per_step': cg['n'], 'avg_launch_ms': round(cg['ms'] / cg['n'], 5),
            'gflop_per_launch': round(cg['flops'] / cg['n'] / 1e9, 3), 'gpu_ms_per_step': round(total, 3),
            'by_kernel': table, 'by_role_ms': {k: round(v[0], 3) for k, v in sorted(roles.items(), key=lambda kv: -kv[1][0])}}


def cpu_train_baseline(cfg, budget_s, size=608):
    """Eager fp32 training step of the same cfg on the host cores (this package's eager modules, bit-equal to the reference's;
    BASELINE.md 3: batch 2, per image; thread count: host_threads()).  The first step (thread-pool start, page-in) is a
    warm-up unless it alone exhausts the budget."""
    from models import Darknet
    from utils.utils import compute_loss
    threads, usable, probe = host_threads()
    torch.manual_seed(0)
    model = Darknet(cfg, (size, size)).train()
    model.nc, model.hyp, model.gr = 80, HYP, 1.0
    x = torch.rand(2, 3, size, size)
    targets = synthetic_labels(2, 8, 80, 7, 'cpu')
    opt = torch.optim.SGD(model.parameters(), lr=HYP['lr0'] * 0.01, momentum=HYP['momentum'], nesterov=True)
    def one():
        pred, _ = model(x)
        loss, _ = compute_loss(pred, targets, model)
        opt.zero_grad()
        loss.backward()
        opt.step()

    t0 = time.time()
    one()
    warm = time.time() - t0
    n, dt = 2, warm
    if warm < budget_s:      # otherwise the warm-up step IS the bounded sample
        t0, n = time.time(), 0
        while True:
            one()
            n += 2
            dt = time.time() - t0
            if dt >= budget_s or n >= 32:
                break
    return dict(value=round(n / dt, 3), unit='images/s', cores=torch.get_num_threads(), kind='port',
                sample='%d images of %s, batch 2, fp32 eager forward + loss + backward + SGD step, %.1f s on %d threads (host: %d CPUs, '
                       '%d usable; conv probe it/s by thread count %s)' % (n, model_label(cfg, size), dt, threads, os.cpu_count() or 1, usable, probe))


def self_launch(n):
    """Re-run this command as `n` ranks under torch.distributed.run on this node; returns the launcher's exit code."""
    import subprocess
    # c10d rendezvous on 127.0.0.1 with port 0: the launcher's store binds a free port itself and hands it to the ranks as
    # MASTER_PORT (no bind-then-close probe that another process could race for)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % n, '--rdzv-backend', 'c10d',
           '--rdzv-endpoint', '127.0.0.1:0', '--local-addr', '127.0.0.1', os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get('OMP_NUM_THREADS', '8'))
    return subprocess.call(cmd, env=env)


def dry_main(args, world, rank):
    """--dry-run: the N-rank contract without a GPU (gloo): barrier-bracketed timing, MAX over ranks, rank 0 prints."""
    from engine import distutil
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo')
    for _ in range(args.warmup):
        time.sleep(0.001)
    elapsed = distutil.timed_region(lambda: time.sleep(0.002 * (rank + 1)), args.steps, dist, torch.device('cpu'))
    if rank == 0:
        print(json.dumps({'metric': 'dry run (no kernels)', 'value': round(world * args.batch * args.steps / elapsed, 2),
                          'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'none', 'data': 'none', 'dry_run': True,
                          'config': {'workload': 'launcher contract only', 'global_batch': world * args.batch,
                                     'parallelism': 'dp%d' % world}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='frames per GPU per step (BASELINE: batch 64/GPU)')
    ap.add_argument('--size', type=int, default=608)
    ap.add_argument('--precision', default='fp16', choices=['fp16', 'fp32', 'int8'])
    ap.add_argument('--cfg', default=os.path.join(PKG, 'cfg', 'yolov3', 'yolov3.cfg'))
    ap.add_argument('--mode', default='both', choices=['both', 'detect', 'train'],
                    help='train: forward + loss + backward + SGD step (BASELINE metric "train fp16", configs[2] at N GPUs); '
                         'detect: forward + NMS (configs[1]); both (default): train is the headline value, detect rides along')
    ap.add_argument('--no-nms', action='store_true')
    ap.add_argument('--two-pass-nms', action='store_true', help='detect leg: model(x) then non_max_suppression(inf) (rounds 1 - 4) instead of model.hip_detect')
    ap.add_argument('--raw-heads', action='store_true', help='keep the random head weights (no NMS candidates at conf 0.3): forward-only timing')
    ap.add_argument('--nms-candidates', type=int, default=100, help='objectness candidates per image the synthetic heads are calibrated to')
    ap.add_argument('--families', action='store_true',
                    help='add the YOLOv3-Mobilenetv3-416 (BASELINE configs[4]) and YOLOv4-608 training steps as rider legs (off by default: the '
                         'default line and its rocprofv3 companion stay those of the YOLOv3-608 legs)')
    ap.add_argument('--no-v4', action='store_true', help='skip the YOLOv4-640 fp16 / int8 rider legs of the default run')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--grad-compress', default='none', choices=['none', 'fp16'],
                    help='DDP communication hook: fp16 = gradient buckets all-reduced in fp16 (half the xGMI bytes), fp32 master grads')
    ap.add_argument('--bucket-mb', type=int, default=25, help='DDP bucket size; the backward runs in 8 ranges of ~31 MB of gradients each')
    ap.add_argument('--share-gpu', action='store_true',
                    help='let ranks share GPUs when fewer than --gpus are visible (smoke runs of the N > 1 path on a 1-GPU box; '
                         'RCCL itself refuses two ranks on one device, so combine with --dist-backend gloo there)')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'], help='nccl = RCCL over xGMI (the product path)')
    ap.add_argument('--dry-run', action='store_true',
                    help='launcher / rendezvous / timing contract only, on CPU over gloo: no kernels run, the value is meaningless')
    args = ap.parse_args()

    if args.gpus > 1 and 'RANK' not in os.environ:
        # called the way the driver calls it (`python bench.py --gpus N`): start one rank per GPU ourselves, the same
        # way `python -m torch.distributed.run --nproc-per-node N bench.py ...` would (reference train.py:94-107 is
        # started by torch.distributed.launch); rank 0's JSON line is the only thing printed on stdout
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d' % (args.gpus, world, args.gpus))
    if args.dry_run:
        return dry_main(args, world, rank)
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    ndev = torch.cuda.device_count()
    if local_rank >= ndev and not args.share_gpu:
        raise SystemExit('rank %d has no GPU of its own (%d visible); --share-gpu lets ranks share a device (smoke runs only)'
                         % (local_rank, ndev))
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    local_rank = dev_index
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:   # gloo moves the gradient buckets through host memory: DDP smoke runs on a shared GPU only
            dist.init_process_group(args.dist_backend)

    if args.precision == 'int8' and args.mode == 'both':
        args.mode = 'detect'   # the int8 PTQ graph is an inference path
    out = None
    if args.mode in ('train', 'both'):
        try:
            out = train_main(args, device, dist, world, rank, local_rank)
        except (NotImplementedError, RuntimeError, MemoryError) as e:  # keep the contract: one JSON line, say what happened
            if args.mode == 'train':
                raise
            train_error = '%s: %s' % (type(e).__name__, str(e)[:300])
            torch.cuda.empty_cache()
            det = detect_main(args, device, dist, world, rank, cpu_baseline_leg=True)
            if rank == 0:
                det['train_error'] = train_error
                print(json.dumps(det))
            if dist is not None:
                dist.destroy_process_group()
            return
    if args.mode in ('detect', 'both'):
        keys = ('metric', 'value', 'unit', 'ms_per_step', 'dtype', 'config', 'roofline_net', 'roofline', 'nms', 'nms_test_settings')

        def rider(name, rargs, eval_nms=False):
            """A detection leg that rides on the headline line; detect_main raises on every rank together, so catching here
            keeps the ranks in step."""
            try:
                d = detect_main(rargs, device, dist, world, rank, cpu_baseline_leg=False, eval_nms=eval_nms)
                if rank == 0 and d is not None:
                    out[name] = {k: d[k] for k in keys if k in d}
            except Exception as e:   # noqa: BLE001 - the rider must not take the headline line down with it
                if rank == 0:
                    out[name + '_error'] = '%s: %s' % (type(e).__name__, str(e)[:300])

        if args.mode == 'detect':
            det = detect_main(args, device, dist, world, rank, cpu_baseline_leg=True)
            if rank == 0:
                out = det
        else:
            rider('detect', args, eval_nms=True)      # the second headline metric of BASELINE.json, measured in the same run
        if args.mode == 'both' and args.precision == 'fp16':
            import copy
            # third headline metric ("detect int8 FPS"): the COS-PTQ graph on the MFMA-i8 engine, same frames, same NMS
            iargs = copy.copy(args)
            iargs.precision = 'int8'
            rider('detect_int8', iargs)
            if not args.no_v4 and os.path.basename(args.cfg) == 'yolov3.cfg':
                # BASELINE.json configs[3]: YOLOv4 640 (CSPDarknet53 + Mish + SPP/PAN) on the int8 PTQ path, batch 32, with its
                # fp16 twin beside it
                for prec, name in (('fp16', 'detect_v4_640'), ('int8', 'detect_int8_v4_640')):
                    vargs = copy.copy(args)
                    vargs.cfg = os.path.join(PKG, 'cfg', 'yolov4', 'yolov4.cfg')
                    vargs.size, vargs.batch, vargs.precision = 640, min(args.batch, 32), prec
                    rider(name, vargs)
    if args.mode == 'both' and args.precision == 'fp16' and world == 1 and args.families \
            and os.path.basename(args.cfg) == 'yolov3.cfg' and out is not None:
        # the other training configurations the reference is used for, on the same step (riders; the headline stays YOLOv3-608):
        # BASELINE.json configs[4] - YOLOv3-Mobilenetv3 (depthwise + squeeze-excite backbone) at 416, batch 64 - and YOLOv4-608, batch 32
        import copy
        for rel, size, batch, name in (('yolov3-mobilenet/yolov3-mobilenet-coco.cfg', 416, 64, 'train_mobilenet_416'),
                                       ('yolov4/yolov4.cfg', 608, 32, 'train_v4_608')):
            targs = copy.copy(args)
            targs.cfg, targs.size, targs.batch = os.path.join(PKG, 'cfg', *rel.split('/')), size, min(args.batch, batch)
            targs.steps, targs.warmup, targs.no_cpu_baseline = min(args.steps, 10), min(args.warmup, 3), True
            try:
                t = train_main(targs, device, dist, world, rank, local_rank)
                r = t.get('roofline') or {}
                out[name] = {k: t[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'dtype', 'config') if k in t}
                out[name]['roofline'] = {k: r[k] for k in ('gpu_ms_per_step', 'batchnorm_passes_ms', 'by_role_ms', 'top_classes') if k in r}
            except Exception as e:   # noqa: BLE001 - a rider must not take the headline line down with it
                out[name + '_error'] = '%s: %s' % (type(e).__name__, str(e)[:300])
                torch.cuda.empty_cache()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def all_ranks_ok(ok, dist, device):
    """True on every rank only if ``ok`` is true on every rank (one MIN all-reduce; no-op for a single process)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def nms_leg(inf, steps):
    """NMS alone on one decoded batch (HIP events on the launch stream): ms per call, candidates and survivors per image."""
    from engine import hiplib
    from utils.utils import non_max_suppression
    lib = hiplib.load()
    n, rows, no = inf.shape
    count = torch.zeros(n, dtype=torch.int32, device=inf.device)
    hiplib.check(lib.yh_nms_candidates(hiplib.ptr(inf), n, rows, no - 5, 0.3, 0, None, None, hiplib.ptr(count), 0,
                                       hiplib.stream_ptr()), 'nms count')
    det = non_max_suppression(inf, conf_thres=0.3, iou_thres=0.6, multi_label=False)   # also warms the candidate bound
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        non_max_suppression(inf, conf_thres=0.3, iou_thres=0.6, multi_label=False)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    kept = [0 if d is None else int(d.shape[0]) for d in det]
    return {'ms_per_call': round(wall, 4), 'gpu_ms_per_call': round(e0.elapsed_time(e1) / steps, 4),
            'candidates_per_image': round(float(count.float().mean()), 1), 'max_candidates': int(count.max()),
            'detections_per_image': round(sum(kept) / max(len(kept), 1), 1), 'settings': 'conf 0.3, iou 0.6, best class, merge'}


def nms_test_settings_leg(inf, images=16, steps=3, size=608):
    """NMS at test.py's settings (reference test.py:15-16, 91: conf 0.001, iou 0.6, multi-label) on `images` frames of the decoded
    batch: the evaluation path works on 10^3 - 10^5 candidates per image where detect.py sees ~10^2.  Reported next to
    the forward time of the same frames so that the O(m^2) stages (rank sort, IoU bit mask) can be judged against it."""
    from engine import hiplib
    from utils.utils import non_max_suppression
    lib = hiplib.load()
    sub = inf[:images].clone()
    # a trained detector's statistics at conf 0.001: a few hundred rows per image carry objectness, most of their class scores pass
    # (COCO evaluation: 10^3 - 3 x 10^4 candidates per image).  The synthetic heads put logit(0.3) at the 100th row and would pass
    # every row x class at 0.001 (1.8 M candidates per image - more than the reference's torchvision NMS could mask either)
    obj = sub[..., 4]
    kth = obj.topk(min(300, obj.shape[1]), dim=1).values[:, -1:]
    sub[..., 4] = torch.where(obj >= kth, obj, torch.zeros_like(obj))
    n, rows, no = sub.shape
    # ... and a trained detector fires on several neighbouring cells / anchors per object: the rows that carry objectness are moved onto
    # 40 objects per image (centre jitter 2 %, size jitter 5 % of the object), so that the greedy scan has clusters to suppress as on
    # COCO (isolated random boxes never suppress anything)
    g = torch.Generator().manual_seed(7)
    k_obj, live = 40, min(300, rows)
    idx = obj.topk(live, dim=1).indices.cpu()
    centers = torch.rand(n, k_obj, 2, generator=g) * size
    sizes = (torch.rand(n, k_obj, 2, generator=g) * 0.3 + 0.04) * size
    which = torch.randint(0, k_obj, (n, live), generator=g)
    pick = which.unsqueeze(-1).expand(-1, -1, 2)
    boxes = torch.cat((centers.gather(1, pick) + torch.randn(n, live, 2, generator=g) * 0.02 * sizes.gather(1, pick),
                       sizes.gather(1, pick) * (1 + torch.randn(n, live, 2, generator=g) * 0.05)), 2)
    sub[torch.arange(n).unsqueeze(1), idx.to(sub.device), :4] = boxes.to(sub.device)
    count = torch.zeros(n, dtype=torch.int32, device=sub.device)
    hiplib.check(lib.yh_nms_candidates(hiplib.ptr(sub), n, rows, no - 5, 0.001, 1, None, None, hiplib.ptr(count), 0,
                                       hiplib.stream_ptr()), 'nms count')
    mmax = int(count.max())
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    det = non_max_suppression(sub, conf_thres=0.001, iou_thres=0.6, multi_label=True)       # warms the candidate bound
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        non_max_suppression(sub, conf_thres=0.001, iou_thres=0.6, multi_label=True)
    e1.record()
    torch.cuda.synchronize()
    kept = [0 if d is None else int(d.shape[0]) for d in det]
    return {'images': n, 'gpu_ms_per_call': round(e0.elapsed_time(e1) / steps, 3), 'candidates_per_image': round(float(count.float().mean()), 1),
            'max_candidates': mmax, 'detections_per_image': round(sum(kept) / max(len(kept), 1), 1),
            'peak_mb': round((torch.cuda.max_memory_allocated() - base) / 1e6, 1),
            'settings': 'conf 0.001, iou 0.6, multi-label (test.py); 300 rows with objectness on 40 objects per image'}


def detect_main(args, device, dist, world, rank, cpu_baseline_leg=True, eval_nms=None):
    """configs[1] / configs[3]: forward + NMS on a resident synthetic batch; returns the JSON dict on rank 0 (None elsewhere).

    The model build (the step that can fail: lowering, memory) happens before any collective and is agreed on by all ranks, so
    a rank that fails raises on EVERY rank instead of leaving the others in the timed region's barrier."""
    from utils.utils import non_max_suppression
    from tools.synthetic_heads import detector_like_heads_
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.rand(args.batch, 3, args.size, args.size, generator=g).to(device)  # resident in HBM before timing
    model, heads, err = None, None, None
    try:
        # a random-weight detector hands NMS nothing at conf 0.3: give the three head convs a trained head's statistics
        # (tools/synthetic_heads.py; every other layer untouched), measured on the first frames of the bench batch
        fm = build_model(args.cfg, args.size, 'fp16' if args.precision == 'int8' else args.precision, device)
        if not args.raw_heads:
            heads = detector_like_heads_(fm, x[:min(16, args.batch)], per_image=args.nms_candidates)
        if args.precision == 'int8':
            calib = 'synthetic power-of-two state covering the float ranges (tools/synthetic_ptq.py)'
            model = None
            if not args.raw_heads and os.environ.get('YOLO_BENCH_PTQ', 'device') == 'device':
                try:
                    model = build_qmodel_calibrated(args.cfg, args.size, device, fm, x[:min(6, args.batch)])
                    calib = 'COS-PTQ calibrated on the device, 3 batches of 2 bench frames (engine/calib.py)'
                except Exception as e:      # noqa: BLE001 - timing leg: fall back to the synthetic state, say so
                    calib += '; device calibration failed: %s: %s' % (type(e).__name__, str(e)[:120])
                    model = None
            if model is None:
                model = build_qmodel_synthetic(args.cfg, args.size, device, float_model=fm, frames=None if args.raw_heads else x[:2])
            heads = dict(heads or {}, int8_state=calib)
        else:
            model = fm
        del fm
        model.hip_return_raw = False      # as detect.py: it reads model(img)[0] only, the raw head maps are not copied out
        with torch.no_grad():
            model(x)                      # builds the plan: lowering errors surface here
        torch.cuda.synchronize()
    except Exception as e:                # noqa: BLE001 - reported, then raised on every rank together
        err = e
    if not all_ranks_ok(err is None, dist, device):
        raise err if err is not None else RuntimeError('detect leg failed on another rank')

    def step():      # detect.py's loop body (this package's detect.py: model.hip_detect = forward + NMS as one engine call)
        with torch.no_grad():
            if args.no_nms:
                model(x)
            elif args.two_pass_nms:
                inf, _, _ = model(x)
                non_max_suppression(inf, conf_thres=0.3, iou_thres=0.6, multi_label=False)
            else:
                model.hip_detect(x, conf_thres=0.3, iou_thres=0.6, multi_label=False)

    from engine import distutil
    for _ in range(args.warmup):
        step()
    # barrier + synchronize on both sides of exactly K steps; MAX over ranks (engine/distutil.py)
    elapsed = distutil.timed_region(step, args.steps, dist, device)

    if rank == 0:
        images = world * args.batch * args.steps
        value = images / elapsed
        flops = conv_flops(model.__dict__['_hip_engine']._plans[tuple(x.shape)])
        gflop_img = sum(flops.values()) / args.batch / 1e9
        peak = PEAK_TFLOPS[args.precision]
        net_tflops = value / world * gflop_img / 1e3
        label = model_label(args.cfg, args.size)
        out = {
            'metric': 'images/sec %s detect %s (%s)' % (label, args.precision, 'forward only' if args.no_nms else 'forward + NMS'),
            'value': round(value, 2), 'unit': 'images/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': {'fp16': 'f16', 'fp32': 'f32', 'int8': 'i8'}[args.precision], 'data': 'synthetic',
            'config': {'workload': '%s %d COCO (80 classes) %s inference, batch %d/GPU, %s'
                                   % (model_family(args.cfg), args.size, args.precision, args.batch,
                                      'forward only' if args.no_nms else 'forward + NMS conf 0.3 iou 0.6'),
                       'global_batch': world * args.batch, 'parallelism': 'replicas x%d' % world,
                       'gflop_per_image': round(gflop_img, 3),
                       'heads': 'random weights, default head bias (NMS sees no candidates)' if heads is None else
                                dict(heads, note='head convs rescaled to a trained detector\'s objectness statistics, tools/synthetic_heads.py')},
            'roofline_net': {'bound': 'mfma', 'achieved': round(net_tflops, 2), 'peak': peak, 'unit': 'TFLOP/s',
                             'frac': round(net_tflops / peak, 4), 'per': 'GPU, whole step incl. NMS and host gaps'},
        }
        if not args.no_nms:
            with torch.no_grad():
                inf, _, _ = model(x)
            out['nms'] = nms_leg(inf, max(3, min(args.steps, 10)))
            if args.precision == 'fp16' and (cpu_baseline_leg if eval_nms is None else eval_nms):      # once per line: the fp16 leg of the headline net
                try:
                    out['nms_test_settings'] = nms_test_settings_leg(inf, size=args.size)
                except Exception as e:       # a rider never takes the line down
                    out['nms_test_settings'] = {'error': '%s: %s' % (type(e).__name__, e)}
        out['roofline'] = roofline_leg(model, x, max(3, min(args.steps, 10)), args.precision)
        if world == 1 and not args.no_cpu_baseline and cpu_baseline_leg:
            out['cpu_baseline'] = cpu_baseline(args.cfg, args.size, args.cpu_seconds)
        else:
            out['cpu_baseline'] = None
    else:
        out = None
    distutil.barrier(dist)
    del model
    torch.cuda.empty_cache()
    return out


if __name__ == '__main__':
    main()
