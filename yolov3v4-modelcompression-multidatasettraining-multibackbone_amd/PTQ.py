"""Post-training quantisation entry point (reference PTQ.py:12-131): evaluate the float model, calibrate the COS-PTQ graph
with train-mode forwards over the calibration split (no gradients), evaluate it, save ``weights/PTQ.pt``.

    python PTQ.py --cfg cfg/yolov3/yolov3.cfg --data data/coco2014.data --weights weights/yolov3.weights

The calibration modules are this package's ``utils/quantized/quantized_ptq_cos.py`` (bit-identical decisions to the
reference's, tests/test_ptq_calibration.py), so networks with max-pools (YOLOv4, the tiny nets) calibrate too.  On a GPU
the calibrated model evaluates on the int8 MFMA engine (``Darknet.forward`` in eval mode, quantized == 3)."""
import argparse
import glob
import os

import torch
from torch.utils.data import DataLoader

import test as test_module
from models import Darknet, attempt_download, load_darknet_weights
from utils import torch_utils
from utils.datasets import LoadImagesAndLabels
from utils.parse_config import parse_data_cfg

wdir = 'weights' + os.sep
PTQ_weights = wdir + 'PTQ.pt'


def calibrate(q_model, batches, device, maxabsscaler=False, augment=False, progress=None):
    """Train-mode forwards without gradients (PTQ.py:76-88): every quantiser votes on its scale once per batch."""
    q_model.train()
    n = 0
    for imgs, _, _, _ in (progress(batches) if progress else batches):
        imgs = imgs.to(device).float() / 256.0
        if maxabsscaler:
            imgs = imgs * 2 - 1
        with torch.no_grad():
            q_model(imgs, augment=augment)
        n += 1
    return n


def PTQ(opt):
    device = torch_utils.select_device(opt.device, batch_size=opt.batch_size)
    print('PTQ only support for one gpu!\n')
    common = dict(is_gray_scale=opt.gray_scale, maxabsscaler=opt.maxabsscaler)
    model = Darknet(opt.cfg, **common)
    q_model = Darknet(opt.cfg, quantized=3, a_bit=opt.a_bit, w_bit=opt.w_bit, shortcut_way=opt.shortcut_way, **common)
    attempt_download(opt.weights)
    if opt.weights.endswith('.pt'):
        state = torch.load(opt.weights, map_location='cpu', weights_only=False)['model']
        model.load_state_dict(state)
        q_model.load_state_dict(state)
    else:
        load_darknet_weights(model, opt.weights)
        load_darknet_weights(q_model, opt.weights, quant=True)
    model.to(device)
    q_model.to(device)

    data = parse_data_cfg(opt.data)
    loaders = {}
    for split, key, subset in (('cali', 'train', opt.subset_len), ('test', 'valid', -1)):
        ds = LoadImagesAndLabels(data[key], opt.img_size, opt.batch_size, rect=True, is_gray_scale=opt.gray_scale, subset_len=subset)
        bs = min(opt.batch_size, len(ds))
        loaders[split] = DataLoader(ds, batch_size=bs, num_workers=min([os.cpu_count(), bs if bs > 1 else 0, 8]),
                                    pin_memory=True, collate_fn=ds.collate_fn)
    evaluate = lambda m, **kw: test_module.test(opt.cfg, data=opt.data, batch_size=opt.batch_size, imgsz=opt.img_size, model=m,
                                                dataloader=loaders['test'], rank=-1, maxabsscaler=opt.maxabsscaler, **kw)
    print('\n<.....................test original model.......................>')
    before = evaluate(model)
    print('\n<.....................Quantize.......................>')
    try:
        from tqdm import tqdm
    except ImportError:
        tqdm = None
    calibrate(q_model, loaders['cali'], device, opt.maxabsscaler, opt.augment, tqdm)
    print('\n<.....................test quantized model.......................>\n')
    after = evaluate(q_model, quantized=3, a_bit=opt.a_bit, w_bit=opt.w_bit)
    os.makedirs(wdir, exist_ok=True)
    torch.save({'epoch': None, 'best_fitness': None, 'training_results': None, 'model': q_model.state_dict(), 'optimizer': None},
               PTQ_weights)
    return before, after


def make_parser():
    parser = argparse.ArgumentParser(prog='PTQ.py')
    parser.add_argument('--cfg', type=str, default='cfg/yolov3-spp.cfg', help='*.cfg path')
    parser.add_argument('--data', type=str, default='data/coco2014.data', help='*.data path')
    parser.add_argument('--weights', type=str, default='weights/yolov3-spp-ultralytics.pt', help='weights path')
    parser.add_argument('--batch-size', type=int, default=16, help='size of each image batch')
    parser.add_argument('--img-size', type=int, default=512, help='inference size (pixels)')
    parser.add_argument('--device', default='', help='device id (i.e. 0 or 0,1) or cpu')
    parser.add_argument('--single-cls', action='store_true', help='train as single-class dataset')
    parser.add_argument('--augment', action='store_true', help='augmented inference')
    parser.add_argument('--a-bit', type=int, default=8, help='a-bit')
    parser.add_argument('--w-bit', type=int, default=8, help='w-bit')
    parser.add_argument('--subset_len', type=int, default=-1, help='calibration set len')
    parser.add_argument('--gray_scale', action='store_true', help='gray scale trainning')
    parser.add_argument('--maxabsscaler', '-mas', action='store_true', help='Standarize input to (-1,1)')
    parser.add_argument('--shortcut_way', type=int, default=1, help='--shortcut quantization way')
    return parser


if __name__ == '__main__':
    opt = make_parser().parse_args()
    for field in ('cfg', 'data'):
        if not os.path.isfile(getattr(opt, field)):
            setattr(opt, field, list(glob.iglob('./**/' + getattr(opt, field), recursive=True))[0])
    print(opt)
    PTQ(opt)
