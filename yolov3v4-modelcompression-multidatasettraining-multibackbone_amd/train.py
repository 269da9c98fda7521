"""Training entry point (reference train.py:52-597): SGD/Adam with the three parameter groups, burn-in, cosine schedule,
gradient accumulation to an effective batch of 64, optional mixed precision (``-mpt``: autocast + GradScaler), EMA,
per-epoch ``test.test`` and ``last.pt`` / ``best.pt`` checkpoints, data-parallel over one process per GPU.

On a GPU the step runs on the HIP training path: ``model(imgs)`` in train mode, ``compute_loss`` and
``loss.backward()`` are the same three calls as in the reference.  Launch for N GPUs of one node with

    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py --cfg ... --data ...

(``LOCAL_RANK`` / ``WORLD_SIZE`` from the launcher; ``nccl`` = RCCL on GPUs, ``gloo`` on CPU).  Not carried over:
tensorboard logging, hyper-parameter evolution, cloud buckets, knowledge-distillation strategies 2-5 (1 is kept),
the FenceMask/GridMask augmentations.  BN-gamma sparsity training (``--prune``/``-s``) uses ``utils.prune_utils`` of a
reference checkout when one is importable (utils/__init__.py).
"""
import argparse
import glob
import math
import os
import random
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.optim as optim
import torch.optim.lr_scheduler as lr_scheduler
from torch.utils.data import DataLoader

import test as test_module   # the library function test.test
from models import Darknet, attempt_download, load_darknet_weights
from utils import torch_utils
from utils.datasets import LoadImagesAndLabels
from utils.parse_config import parse_data_cfg
from utils.utils import (compute_loss, compute_lost_KD, fitness, init_seeds, labels_to_class_weights, labels_to_image_weights,
                         plot_images, plot_results, strip_optimizer)

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    tqdm = lambda x, **kw: x

wdir = 'weights' + os.sep
last, best, results_file = wdir + 'last.pt', wdir + 'best.pt', 'results.txt'

# reference train.py:25-42
hyp = {'giou': 3.54, 'cls': 37.4, 'cls_pw': 1.0, 'obj': 64.3, 'obj_pw': 1.0, 'iou_t': 0.20, 'lr0': 0.01, 'lrf': 0.0005,
       'momentum': 0.937, 'weight_decay': 0.0005, 'fl_gamma': 0.0, 'hsv_h': 0.0138, 'hsv_s': 0.678, 'hsv_v': 0.36,
       'degrees': 1.98 * 0, 'translate': 0.05 * 0, 'scale': 0.05 * 0, 'shear': 0.641 * 0}


def _is_main(rank):
    return rank in (-1, 0)


def load_teacher(t_cfg, t_weights, device):
    """The knowledge-distillation teacher with its weights (reference train.py:185-192): .pt checkpoint or darknet .weights;
    anything else is an error (distilling from a randomly initialised teacher is never what was meant)."""
    t_model = Darknet(t_cfg).to(device)
    if t_weights.endswith('.pt'):
        t_model.load_state_dict(torch.load(t_weights, map_location=device, weights_only=False)['model'])
    elif t_weights.endswith('.weights'):
        load_darknet_weights(t_model, t_weights)
    else:
        raise Exception('pls provide proper teacher weights for knowledge distillation')
    return t_model.eval()


def build_optimizer(model, opt, hyp, device):
    """Parameter groups and update rule of reference train.py:120-146: biases / decayed conv weights / the rest (+ learned
    quantiser scales for quantized == 2); nesterov SGD at lr0, or Adam at lr0 * 0.005 for --adam and every quantised graph."""
    pg0, pg1, pg2, pg3 = [], [], [], []
    for k, v in dict(model.named_parameters()).items():
        if '.bias' in k:
            pg2.append(v)
        elif 'Conv2d.weight' in k:
            pg1.append(v)
        elif 'scale' in k and opt.quantized == 2:
            pg3.append(v)
        else:
            pg0.append(v)
    # on a GPU torch's single-launch multi-tensor kernels do the update (with GradScaler's unscale folded in): the default
    # per-op foreach form costs ~5 ms of a 72 ms YOLOv3-608 step
    fused = {'fused': True} if device.type == 'cuda' else {}
    if opt.adam or opt.quantized != -1:
        optimizer = optim.Adam(pg0, lr=hyp['lr0'] * 0.005, **fused)
        if opt.quantized == 2:
            optimizer.add_param_group({'params': pg3})
    else:
        optimizer = optim.SGD(pg0, lr=hyp['lr0'], momentum=hyp['momentum'], nesterov=True, **fused)
    optimizer.add_param_group({'params': pg1, 'weight_decay': hyp['weight_decay']})
    optimizer.add_param_group({'params': pg2})
    return optimizer


def sparsity_layers(prune, module_defs):
    """BN layers under the L1 sparsity penalty for --prune 0 / 1 / 2 (reference train.py:237-262)."""
    from utils.prune_utils import parse_module_defs, parse_module_defs2, parse_module_defs4
    if prune == 0:      # regular prune: convs outside shortcuts
        _, _, prune_idx = parse_module_defs(module_defs)
    elif prune == 1:    # shortcut prune
        _, _, prune_idx, _, _ = parse_module_defs2(module_defs)
    elif prune == 2:    # layer prune
        _, _, prune_idx = parse_module_defs4(module_defs)
    else:
        raise ValueError('--prune must be 0, 1 or 2')
    return prune_idx


def train(opt, hyp):
    cfg, data, epochs, batch_size, weights = opt.cfg, opt.data, opt.epochs, opt.batch_size, opt.weights
    imgsz_min, imgsz_max, imgsz_test = opt.img_size
    gs = 32
    assert imgsz_min % gs == 0, '--img-size %g must be a %g-multiple' % (imgsz_min, gs)
    opt.multi_scale |= imgsz_min != imgsz_max
    if opt.multi_scale:
        if imgsz_min == imgsz_max:
            imgsz_min, imgsz_max = int(imgsz_min // 1.5), int(imgsz_max // 0.667)
        grid_min, grid_max = imgsz_min // gs, imgsz_max // gs
        imgsz_min, imgsz_max = int(grid_min * gs), int(grid_max * gs)
    img_size = imgsz_max

    init_seeds()
    data_dict = parse_data_cfg(data)
    train_path, test_path, nc = data_dict['train'], data_dict['valid'], int(data_dict['classes'])
    hyp = dict(hyp)
    hyp['cls'] *= nc / 80

    # one process per GPU; the launcher provides the rendezvous (train.py:96-110 hard-codes nccl + tcp://127.0.0.1:9999)
    rank, world = opt.local_rank, int(os.environ.get('WORLD_SIZE', '1'))
    device = torch_utils.select_device(opt.device, batch_size=batch_size)
    distributed = rank != -1 and world > 1
    if distributed:
        if device.type != 'cpu':
            torch.cuda.set_device(rank)
            device = torch.device('cuda', rank)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC only on this driver (RCCL)
        dist.init_process_group(backend='nccl' if device.type != 'cpu' else 'gloo', init_method='env://')
        assert batch_size % world == 0, '--batch-size must be multiple of the process count'
        batch_size //= world
    accumulate = max(round(64 / (batch_size * (world if distributed else 1))), 1)
    if _is_main(rank):
        for f in glob.glob('*_batch*.jpg') + glob.glob(results_file):
            os.remove(f)
        os.makedirs(wdir, exist_ok=True)

    steps = math.ceil(len(open(train_path).readlines()) / batch_size) * epochs if os.path.isfile(train_path) else 0
    model = Darknet(cfg, quantized=opt.quantized, a_bit=opt.a_bit, w_bit=opt.w_bit, steps=steps, is_gray_scale=opt.gray_scale,
                    maxabsscaler=opt.maxabsscaler, shortcut_way=opt.shortcut_way).to(device)
    t_model = load_teacher(opt.t_cfg, opt.t_weights, device) if opt.t_cfg else None

    optimizer = build_optimizer(model, opt, hyp, device)

    start_epoch, best_fitness = 0, 0.0
    if weights:
        attempt_download(weights)
        if weights.endswith('.pt'):
            ckpt = torch.load(weights, map_location='cpu', weights_only=False)
            state = {k: v for k, v in ckpt['model'].items() if k in model.state_dict() and model.state_dict()[k].numel() == v.numel()}
            model.load_state_dict(state, strict=False)
            if ckpt.get('optimizer') is not None and opt.resume:
                optimizer.load_state_dict(ckpt['optimizer'])
                best_fitness = ckpt.get('best_fitness', 0.0)
            if ckpt.get('training_results') and _is_main(rank):
                with open(results_file, 'w') as f:
                    f.write(ckpt['training_results'])
            if opt.resume:
                start_epoch = ckpt.get('epoch', -1) + 1
            del ckpt
        else:
            load_darknet_weights(model, weights, pt=opt.pt, quant=(opt.quantized != -1))

    lf = lambda x: (((1 + math.cos(x * math.pi / epochs)) / 2) ** 1.0) * 0.95 + 0.05      # cosine (train.py:198-203)
    if opt.quantized != -1:
        scheduler = lr_scheduler.MultiStepLR(optimizer, milestones=[epochs // 5, epochs // 2, int(epochs // 1.25)], gamma=0.3)
    else:
        scheduler = lr_scheduler.LambdaLR(optimizer, lr_lambda=lf)
    scheduler.last_epoch = start_epoch - 1

    core = model
    if distributed:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[rank] if device.type != 'cpu' else None,
                                                          output_device=rank if device.type != 'cpu' else None)
        if opt.grad_compress == 'fp16':   # gradient buckets cross xGMI as fp16 (half the bytes), master gradients stay fp32
            from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
            model.register_comm_hook(None, default_hooks.fp16_compress_hook)
        model.yolo_layers = core.yolo_layers

    # Device augmentation (the default on a GPU; --host-augment restores the host loader): items arrive as recipes (cropped source
    # frames + geometry + gains) and one HIP kernel per item does the mosaic, warp, HSV, flip, transpose and /256 on the GPU
    # (engine/preprocess.py render_mosaic_items, csrc/augment.hip) - bit-identical to the host items (tests/test_augment.py), at
    # 0.2 ms instead of 49 ms of a host core per item, which is what an 8-GPU node needs to be fed
    # --rect batches are letterboxed per batch rectangle; the device recipe covers only rectangles that need no second resize
    # (utils/datasets.py rect_train_item), so rect training keeps the host loader unless --device-augment asks for the device path
    # explicitly (the default must not turn a working --rect run into a NotImplementedError)
    device_augment = device.type == 'cuda' and not getattr(opt, 'host_augment', False) and \
        (not opt.rect or getattr(opt, 'device_augment', False))
    dataset = LoadImagesAndLabels(train_path, img_size, batch_size, augment=True, hyp=hyp, rect=opt.rect, cache_images=opt.cache_images,
                                  rank=rank, is_gray_scale=opt.gray_scale, device_augment=device_augment,
                                  arith=getattr(opt, 'image_arith', None))
    nw = min([os.cpu_count() or 1, batch_size if batch_size > 1 else 0, 8])
    sampler = torch.utils.data.distributed.DistributedSampler(dataset) if distributed else None
    dataloader = DataLoader(dataset, batch_size=batch_size, num_workers=nw, shuffle=(sampler is None and not opt.rect), sampler=sampler,
                            pin_memory=device.type != 'cpu', collate_fn=dataset.collate_fn)
    testloader = None
    if _is_main(rank) or distributed:
        testset = LoadImagesAndLabels(test_path, imgsz_test, max(batch_size // 4, 1), hyp=hyp, rect=True, rank=rank,
                                      is_gray_scale=opt.gray_scale, device_letterbox=device.type == 'cuda',
                                      arith=getattr(opt, 'image_arith', None) if device.type == 'cuda' else None)
        testloader = DataLoader(testset, batch_size=max(batch_size // 4, 1), num_workers=nw, pin_memory=device.type != 'cpu',
                                collate_fn=testset.collate_fn)

    for m in {model, core}:
        m.nc, m.hyp, m.gr = nc, hyp, 1.0
        m.class_weights = labels_to_class_weights(dataset.labels, nc).to(device)
    ema = torch_utils.ModelEMA(core) if opt.ema else None

    prune_idx = None
    if opt.prune != -1:   # BN-gamma sparsity (network slimming) needs the reference's utils.prune_utils
        prune_idx = sparsity_layers(opt.prune, core.module_defs)
        from utils.prune_utils import BNOptimizer

    nb = len(dataloader)
    n_burn = max(3 * nb, 500)
    maps = np.zeros(nc)
    results = (0, 0, 0, 0, 0, 0, 0)
    scaler = torch.amp.GradScaler('cuda', enabled=opt.mpt and device.type != 'cpu')
    t0 = time.time()
    if _is_main(rank):
        print('Image sizes %g - %g train, %g test' % (imgsz_min, imgsz_max, imgsz_test))
        print('Using %g dataloader workers' % nw)
        print('Starting training for %g epochs...' % epochs)
    for epoch in range(start_epoch, epochs):
        if sampler is not None:
            sampler.set_epoch(epoch)
        model.train()
        if dataset.image_weights:
            w = core.class_weights.cpu().numpy() * (1 - maps) ** 2
            image_weights = labels_to_image_weights(dataset.labels, nc=nc, class_weights=w)
            dataset.indices = random.choices(range(dataset.n), weights=image_weights, k=dataset.n)
        mloss = torch.zeros(4, device=device)
        if _is_main(rank):
            print(('\n' + '%10s' * 8) % ('Epoch', 'gpu_mem', 'GIoU', 'obj', 'cls', 'total', 'targets', 'img_size'))
        pbar = enumerate(dataloader)
        if _is_main(rank):
            pbar = tqdm(pbar, total=nb)
        for i, (imgs, targets, paths, _) in pbar:
            ni = i + nb * epoch
            if not torch.is_tensor(imgs):                # recipes (MosaicBatch): rendered on the device, bit-identical to the host items / 256
                from engine.preprocess import render_mosaic_items
                imgs = render_mosaic_items(imgs, device, dtype=torch.float32, divisor=256.0)
            else:
                imgs = imgs.to(device).float() / 256.0   # uint8 -> [0, 1) (train.py:346-350)
            if opt.maxabsscaler:
                imgs = imgs * 2 - 1
            targets = targets.to(device)

            if ni <= n_burn and opt.quantized == -1:     # burn-in: lr / momentum / giou ratio / accumulation ramp
                xi = [0, n_burn]
                for m in {model, core}:
                    m.gr = float(np.interp(ni, xi, [0.0, 1.0]))
                accumulate = max(1, int(np.interp(ni, xi, [1, 64 / (batch_size * (world if distributed else 1))]).round()))
                for j, g in enumerate(optimizer.param_groups):
                    g['lr'] = float(np.interp(ni, xi, [0.1 if j == 2 else 0.0, g['initial_lr'] * lf(epoch)]))
                    g['weight_decay'] = float(np.interp(ni, xi, [0.0, hyp['weight_decay'] if j == 1 else 0.0]))
                    if 'momentum' in g:
                        g['momentum'] = float(np.interp(ni, xi, [0.9, hyp['momentum']]))

            if opt.multi_scale:
                if ni / accumulate % 1 == 0:
                    img_size = random.randrange(grid_min, grid_max + 1) * gs
                sf = img_size / max(imgs.shape[2:])
                if sf != 1:
                    ns = [math.ceil(x * sf / gs) * gs for x in imgs.shape[2:]]
                    imgs = torch.nn.functional.interpolate(imgs, size=ns, mode='bilinear', align_corners=False)

            with torch.autocast(device.type if device.type != 'cpu' else 'cpu', dtype=torch.float16,
                                enabled=opt.mpt and device.type != 'cpu'):
                pred, feature_s = model(imgs)
            loss, loss_items = compute_loss([p.float() for p in pred], targets, model)
            if not torch.isfinite(loss):
                print('WARNING: non-finite loss, ending training ', loss_items)
                return results
            if t_model is not None:
                with torch.no_grad():
                    _, output_t, _ = t_model(imgs)
                if opt.KDstr != 1:
                    raise NotImplementedError('only knowledge-distillation strategy 1 (soft targets) is carried over')
                loss = loss + compute_lost_KD(pred, output_t, nc, imgs.size(0))

            loss = loss * (batch_size * (world if distributed else 1) / 64)
            scaler.scale(loss).backward()
            if prune_idx is not None:
                BNOptimizer.updateBN(True, core.module_list, opt.s, prune_idx)
            if ni % accumulate == 0:
                scaler.step(optimizer)
                scaler.update()
                optimizer.zero_grad()
                if ema is not None:
                    ema.update(core)

            mloss = (mloss * i + loss_items.to(device)) / (i + 1)
            if _is_main(rank):
                mem = '%.3gG' % (torch.cuda.memory_reserved() / 1E9 if torch.cuda.is_available() else 0)
                line = ('%10s' * 2 + '%10.3g' * 6) % ('%g/%g' % (epoch, epochs - 1), mem, *mloss.tolist(), len(targets), img_size)
                if hasattr(pbar, 'set_description'):
                    pbar.set_description(line)
                if ni < 1:
                    os.makedirs('train_sample', exist_ok=True)
                    plot_images(images=imgs, targets=targets, paths=paths, fname='train_sample/train_batch%g.jpg' % ni,
                                is_gray_scale=opt.gray_scale)

        scheduler.step()
        if ema is not None:
            ema.update_attr(core)
        final_epoch = epoch + 1 == epochs
        if (not opt.notest or final_epoch) and testloader is not None:
            results, maps = test_module.test(cfg, data_dict, batch_size=max(batch_size // 4, 1), imgsz=imgsz_test,
                                             model=ema.ema if ema is not None else core, save_json=False, dataloader=testloader,
                                             multi_label=ni > n_burn, quantized=opt.quantized, a_bit=opt.a_bit, w_bit=opt.w_bit,
                                             rank=rank, plot=False, is_gray_scale=opt.gray_scale, maxabsscaler=opt.maxabsscaler,
                                             shortcut_way=opt.shortcut_way)
        if _is_main(rank):
            with open(results_file, 'a') as f:
                f.write(line + '%10.3g' * 7 % tuple(results) + '\n')
            fi = float(fitness(np.array(results).reshape(1, -1))[0])
            if fi > best_fitness:
                best_fitness = fi
            if not opt.nosave or final_epoch:
                with open(results_file) as f:
                    ckpt = {'epoch': epoch, 'best_fitness': best_fitness, 'training_results': f.read(),
                            'model': (ema.ema if ema is not None else core).state_dict(),
                            'optimizer': None if final_epoch else optimizer.state_dict()}
                torch.save(ckpt, last)
                if best_fitness == fi and not final_epoch:
                    torch.save(ckpt, best)
                del ckpt

    if _is_main(rank):
        if opt.name:
            for a, b in ((results_file, 'results%s.txt' % ('_' + opt.name)), (last, wdir + 'last_%s.pt' % opt.name),
                         (best, wdir + 'best_%s.pt' % opt.name)):
                if os.path.exists(a):
                    os.replace(a, b)
                    if b.endswith('.pt'):
                        strip_optimizer(b)
        plot_results()
        print('%g epochs completed in %.3f hours.\n' % (epochs - start_epoch, (time.time() - t0) / 3600))
    if distributed:
        dist.destroy_process_group()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()
    return results


def make_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--epochs', type=int, default=300)
    parser.add_argument('--batch-size', type=int, default=16)   # effective batch = batch_size * accumulate = 64
    parser.add_argument('--cfg', type=str, default='cfg/yolov3/yolov3.cfg', help='*.cfg path')
    parser.add_argument('--t_cfg', type=str, default='', help='teacher model cfg file path for knowledge distillation')
    parser.add_argument('--data', type=str, default='data/coco2017.data', help='*.data path')
    parser.add_argument('--multi-scale', action='store_true', help='adjust (67%% - 150%%) img_size every 10 batches')
    parser.add_argument('--img-size', nargs='+', type=int, default=[320, 640], help='[min_train, max-train, test]')
    parser.add_argument('--rect', action='store_true', help='rectangular training')
    parser.add_argument('--resume', action='store_true', help='resume training from last.pt')
    parser.add_argument('--nosave', action='store_true', help='only save final checkpoint')
    parser.add_argument('--notest', action='store_true', help='only test final epoch')
    parser.add_argument('--cache-images', action='store_true', help='cache images for faster training')
    parser.add_argument('--device-augment', action='store_true',
                        help='(default on a GPU; kept for command lines of earlier rounds) mosaic / affine / HSV / flip of the training '
                             'items on the GPU: same random streams, same pixels as the host loader')
    parser.add_argument('--host-augment', action='store_true', help='augment the training items on the host cores even when training on a GPU')
    parser.add_argument('--image-arith', choices=['pillow', 'cv2'], default=None,
                        help="uint8 arithmetic of the device input pipeline: 'pillow' = this package's host loader, 'cv2' = the reference's "
                             "OpenCV calls restated (GPU only: needs --device-augment); default: $YOLO_IMAGE_ARITH or pillow")
    parser.add_argument('--weights', type=str, default='', help='initial weights path')
    parser.add_argument('--t_weights', type=str, default='', help='teacher model weights')
    parser.add_argument('--KDstr', type=int, default=-1, help='KD strategy')
    parser.add_argument('--name', default='', help='renames results.txt to results_name.txt if supplied')
    parser.add_argument('--device', default='', help='device id (i.e. 0 or 0,1 or cpu)')
    parser.add_argument('--adam', action='store_true', help='use adam optimizer')
    parser.add_argument('--ema', action='store_true', help='use ema')
    parser.add_argument('--pretrain', '-pt', dest='pt', action='store_true', help='load the whole darknet weights file')
    parser.add_argument('--mixedprecision', '-mpt', dest='mpt', action='store_true', help='mixed precision training')
    parser.add_argument('--s', type=float, default=0.001, help='scale sparse rate')
    parser.add_argument('--prune', type=int, default=-1, help='0: normal / regular prune, 1: shortcut prune, 2: layer prune')
    parser.add_argument('--quantized', type=int, default=-1, help='quantization way')
    parser.add_argument('--shortcut_way', type=int, default=1, help='--shortcut quantization way')
    parser.add_argument('--a-bit', type=int, default=8, help='a-bit')
    parser.add_argument('--w-bit', type=int, default=8, help='w-bit')
    parser.add_argument('--gray-scale', action='store_true', help='gray scale training')
    parser.add_argument('--maxabsscaler', '-mas', action='store_true', help='standardise input to (-1, 1)')
    parser.add_argument('--rank', default=0, help='rank of current process')
    parser.add_argument('--grad-compress', default='none', choices=['none', 'fp16'], help='DDP comm hook: all-reduce gradient buckets in fp16')
    parser.add_argument('--local_rank', type=int, default=int(os.environ.get('LOCAL_RANK', '-1')), help='set by the launcher')
    return parser


if __name__ == '__main__':
    opt = make_parser().parse_args()
    if opt.resume and not opt.weights:
        opt.weights = last
    for key in ('cfg', 'data'):
        found = glob.glob('./**/' + getattr(opt, key), recursive=True)
        if found and not os.path.isfile(getattr(opt, key)):
            setattr(opt, key, found[0])
    opt.img_size.extend([opt.img_size[-1]] * (3 - len(opt.img_size)))
    if _is_main(opt.local_rank):
        print(opt)
    train(opt, hyp)
