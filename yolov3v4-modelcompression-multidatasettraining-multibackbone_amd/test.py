"""Evaluation entry point: ``test(cfg, data, ...)`` -> ``((P, R, mAP@0.5, F1, val GIoU, val obj, val cls), maps)``.

Same library signature and return contract as the reference's ``test.py:10-253`` (train.py calls it every epoch, the
prune scripts call ``test(...)[0][2]`` for mAP), fresh code.  On a CUDA device the model forward and the NMS run on
the HIP path (``models.Darknet`` / ``utils.utils.non_max_suppression``); matching and AP are host side.
"""
import argparse
import glob
import json
import os
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader

from models import Darknet, attempt_download, load_darknet_weights
from utils import torch_utils
from utils.datasets import LoadImagesAndLabels, LetterboxBatch
from utils.parse_config import parse_data_cfg
from utils.utils import (ap_per_class, box_iou, clip_coords, coco80_to_coco91_class, compute_loss, load_classes,
                         non_max_suppression, output_to_target, plot_images, scale_coords, xywh2xyxy, xyxy2xywh)

try:
    from tqdm import tqdm
except Exception:  # pragma: no cover
    tqdm = lambda x, **kw: x

opt = None   # set by the CLI below; test() reads it only when it has to build the model itself (reference test.py:32-33)


def _match(pred, labels, whwh, iouv):
    """Per-prediction correctness flags (n, niou): every label may be claimed once, by a same-class prediction whose IoU
    with it exceeds the threshold, predictions visited per class in descending-IoU-agnostic order like test.py:170-186."""
    correct = torch.zeros(pred.shape[0], iouv.numel(), dtype=torch.bool, device=pred.device)
    if not len(labels):
        return correct
    tcls = labels[:, 0]
    tbox = xywh2xyxy(labels[:, 1:5]) * whwh
    claimed = []
    for cls in torch.unique(tcls):
        ti = (cls == tcls).nonzero(as_tuple=False).view(-1)
        pi = (cls == pred[:, 5]).nonzero(as_tuple=False).view(-1)
        if not pi.shape[0]:
            continue
        ious, best = box_iou(pred[pi, :4], tbox[ti]).max(1)
        for j in (ious > iouv[0]).nonzero(as_tuple=False).view(-1):
            d = int(ti[best[j]])
            if d not in claimed:
                claimed.append(d)
                correct[pi[j]] = ious[j] > iouv
                if len(claimed) == len(labels):
                    break
    return correct


def test(cfg, data, weights=None, batch_size=16, imgsz=416, conf_thres=0.001, iou_thres=0.6, save_json=False, augment=False,
         model=None, dataloader=None, multi_label=True, quantized=-1, a_bit=8, w_bit=8, rank=-1, plot=True, is_gray_scale=False,
         maxabsscaler=False, shortcut_way=-1):
    if model is None:
        device = torch_utils.select_device(opt.device if opt is not None else '', batch_size=batch_size)
        verbose = opt is not None and getattr(opt, 'task', 'test') == 'test'
        for f in glob.glob('test_batch*.jpg'):
            os.remove(f)
        model = Darknet(cfg, imgsz, quantized=quantized, a_bit=a_bit, w_bit=w_bit, is_gray_scale=is_gray_scale,
                        maxabsscaler=maxabsscaler, shortcut_way=shortcut_way)
        if weights:
            attempt_download(weights)
            if weights.endswith('.pt'):
                model.load_state_dict(torch.load(weights, map_location='cpu', weights_only=False)['model'])
            else:
                load_darknet_weights(model, weights, quant=(quantized != -1))
        if quantized == -1:
            model.fuse()
        model.to(device)
    else:   # called by train.py / the prune scripts with a live model
        device = next(model.parameters()).device
        verbose = False

    data = parse_data_cfg(data) if isinstance(data, str) else data
    nc = int(data['classes'])
    names = load_classes(data['names']) if data.get('names') and os.path.isfile(str(data['names'])) else [str(i) for i in range(nc)]
    iouv = torch.linspace(0.5, 0.95, 10).to(device)[0].view(1)   # mAP@0.5 only, like the reference
    niou = iouv.numel()

    if dataloader is None:
        # on a GPU the items are recipes: load_image's resize, letterbox's border, / 256 and HWC -> CHW run there, bit-identical to
        # the host loader's items (engine/preprocess.py render_letterbox_items; --host-letterbox restores the host path)
        dataset = LoadImagesAndLabels(data['valid'], imgsz, batch_size, rect=True, is_gray_scale=is_gray_scale,
                                      device_letterbox=device.type == 'cuda' and not getattr(opt, 'host_letterbox', False),
                                      arith=getattr(opt, 'image_arith', None))
        batch_size = min(batch_size, len(dataset))
        dataloader = DataLoader(dataset, batch_size=batch_size, num_workers=min([os.cpu_count() or 1, batch_size if batch_size > 1 else 0, 8]),
                                pin_memory=device.type != 'cpu', collate_fn=dataset.collate_fn)

    was_training = model.training
    model.eval()
    seen = 0
    coco91 = coco80_to_coco91_class()
    header = ('%20s' + '%10s' * 6) % ('Class', 'Images', 'Targets', 'P', 'R', 'mAP@0.5', 'F1')
    p = r = f1 = mp = mr = map50 = mf1 = t_inf = t_nms = 0.
    fused_calls = 0
    loss = torch.zeros(3, device=device)
    jdict, stats, ap, ap_class = [], [], [], []
    for batch_i, (imgs, targets, paths, shapes) in enumerate(tqdm(dataloader, desc=header) if rank in (-1, 0) else dataloader):
        if isinstance(imgs, LetterboxBatch):
            from engine.preprocess import render_letterbox_items
            imgs = render_letterbox_items(imgs, device, maxabsscaler=maxabsscaler)
        else:
            imgs = imgs.to(device).float() / 256.0      # uint8 -> [0, 1): the reference divides by 256 (test.py:96)
            if maxabsscaler:
                imgs = imgs * 2 - 1
        targets = targets.to(device)
        nb, _, height, width = imgs.shape
        whwh = torch.tensor([width, height, width, height], dtype=torch.float32, device=device)
        with torch.no_grad():
            t = torch_utils.time_synchronized()
            if not augment and not hasattr(model, 'hyp') and hasattr(model, 'hip_detect'):
                # stand-alone evaluation (no validation loss wanted): forward + NMS as one engine call - on the HIP path the decoded
                # (N, rows, 5 + nc) tensor is never written (models.Darknet.hip_detect; same detections bit for bit)
                output = model.hip_detect(imgs, conf_thres, iou_thres, multi_label=multi_label)
                t_inf += torch_utils.time_synchronized() - t
                fused_calls += 1
            else:
                inf_out, train_out = model(imgs, augment=augment)[:2]
                t_inf += torch_utils.time_synchronized() - t
                if hasattr(model, 'hyp') and train_out is not None:
                    loss += compute_loss([x.float() for x in train_out], targets, model)[1][:3].to(device)
                t = torch_utils.time_synchronized()
                output = non_max_suppression(inf_out, conf_thres=conf_thres, iou_thres=iou_thres, multi_label=multi_label)
                t_nms += torch_utils.time_synchronized() - t

        for si, pred in enumerate(output):
            labels = targets[targets[:, 0] == si, 1:]
            tcls = labels[:, 0].tolist() if len(labels) else []
            seen += 1
            if pred is None:
                if len(labels):
                    stats.append((torch.zeros(0, niou, dtype=torch.bool), torch.Tensor(), torch.Tensor(), tcls))
                continue
            clip_coords(pred, (height, width))
            if save_json:
                image_id = Path(paths[si]).stem.split('_')[-1]
                image_id = int(image_id) if image_id.isdigit() else Path(paths[si]).stem
                box = pred[:, :4].clone()
                if shapes[si] is not None:
                    scale_coords(imgs[si].shape[1:], box, shapes[si][0], shapes[si][1])
                box = xyxy2xywh(box)
                box[:, :2] -= box[:, 2:] / 2
                for row, b in zip(pred.tolist(), box.tolist()):
                    cid = int(row[5])
                    jdict.append({'image_id': image_id, 'category_id': coco91[cid] if cid < len(coco91) else cid,
                                  'bbox': [round(x, 3) for x in b], 'score': round(row[4], 5)})
            correct = _match(pred, labels, whwh, iouv)
            stats.append((correct.cpu(), pred[:, 4].cpu(), pred[:, 5].cpu(), tcls))

        if batch_i < 1 and plot:
            plot_images(imgs, targets, paths=paths, names=names, fname='test_batch%g_gt.jpg' % batch_i, is_gray_scale=is_gray_scale)
            plot_images(imgs, output_to_target(output, width, height), paths=paths, names=names,
                        fname='test_batch%g_pred.jpg' % batch_i, is_gray_scale=is_gray_scale)

    stats = [np.concatenate(x, 0) for x in zip(*stats)] if stats else []
    if len(stats) and stats[0].shape[0]:
        p, r, ap, f1, ap_class = ap_per_class(*stats)
        if niou > 1:
            p, r, ap, f1 = p[:, 0], r[:, 0], ap.mean(1), ap[:, 0]
        p, r, ap, f1 = (np.asarray(v).reshape(len(ap_class), -1)[:, 0] for v in (p, r, ap, f1))
        mp, mr, map50, mf1 = p.mean(), r.mean(), ap.mean(), f1.mean()
        nt = np.bincount(stats[3].astype(np.int64), minlength=nc)
    else:
        nt = np.zeros(1)

    row = '%20s' + '%10.3g' * 6
    if rank in (-1, 0):
        print(row % ('all', seen, nt.sum(), mp, mr, map50, mf1))
    if verbose and nc > 1 and len(stats):
        for i, c in enumerate(ap_class):
            print(row % (names[c], seen, nt[c], p[i], r[i], ap[i], f1[i]))
    if (verbose or save_json) and seen:
        ms = tuple(x / seen * 1E3 for x in (t_inf, t_nms, t_inf + t_nms)) + (imgsz, imgsz, batch_size)
        if fused_calls:
            # forward + NMS ran as one engine call: there is no separate NMS time to report (the reference's line,
            # test.py:257, splits them - compare the TOTAL, or run with a model that carries `hyp` for the two-pass path)
            print('Speed: %.1f ms inference+NMS (one fused call) per %gx%g image at batch-size %g' % (ms[2], imgsz, imgsz, batch_size))
        else:
            print('Speed: %.1f/%.1f/%.1f ms inference/NMS/total per %gx%g image at batch-size %g' % ms)
    if save_json and len(jdict):
        with open('results.json', 'w') as f:
            json.dump(jdict, f)
        print('wrote results.json (%d detections); COCO scoring needs pycocotools, which this image does not ship' % len(jdict))

    maps = np.zeros(nc) + map50
    for i, c in enumerate(ap_class):
        maps[c] = ap[i]
    if was_training:
        model.train()
    return (mp, mr, map50, mf1, *(loss.cpu() / max(len(dataloader), 1)).tolist()), maps


if __name__ == '__main__':
    parser = argparse.ArgumentParser(prog='test.py')
    parser.add_argument('--cfg', type=str, default='cfg/yolov3/yolov3.cfg', help='*.cfg path')
    parser.add_argument('--data', type=str, default='data/coco2014.data', help='*.data path')
    parser.add_argument('--weights', type=str, default='', help='weights path (.pt or darknet .weights); empty = random init')
    parser.add_argument('--batch-size', type=int, default=16, help='size of each image batch')
    parser.add_argument('--img-size', type=int, default=512, help='inference size (pixels)')
    parser.add_argument('--conf-thres', type=float, default=0.001, help='object confidence threshold')
    parser.add_argument('--iou-thres', type=float, default=0.6, help='IOU threshold for NMS')
    parser.add_argument('--save-json', action='store_true', help='save a cocoapi-compatible JSON results file')
    parser.add_argument('--task', default='test', help="'test', 'study', 'benchmark'")
    parser.add_argument('--device', default='', help='device id (i.e. 0 or 0,1) or cpu')
    parser.add_argument('--augment', action='store_true', help='augmented inference')
    parser.add_argument('--host-letterbox', action='store_true', help='resize / letterbox the evaluation images on the host (default on a GPU: on the device)')
    parser.add_argument('--image-arith', choices=['pillow', 'cv2'], default=None,
                        help="uint8 arithmetic of the device loader: 'pillow' = the host loader's, 'cv2' = the reference's OpenCV calls restated")
    parser.add_argument('--quantized', type=int, default=-1, help='quantization way')
    parser.add_argument('--shortcut_way', type=int, default=1, help='--shortcut quantization way')
    parser.add_argument('--a-bit', type=int, default=8, help='a-bit')
    parser.add_argument('--w-bit', type=int, default=8, help='w-bit')
    parser.add_argument('--gray-scale', action='store_true', help='gray scale training')
    parser.add_argument('--maxabsscaler', '-mas', action='store_true', help='standardise input to (-1, 1)')
    opt = parser.parse_args()
    opt.save_json = opt.save_json or any(x in opt.data for x in ('coco.data', 'coco2014.data', 'coco2017.data'))
    for key in ('cfg', 'data'):
        found = glob.glob('./**/' + getattr(opt, key), recursive=True)
        if found and not os.path.isfile(getattr(opt, key)):
            setattr(opt, key, found[0])
    print(opt)
    if opt.task == 'test':
        test(opt.cfg, opt.data, opt.weights, opt.batch_size, opt.img_size, opt.conf_thres, opt.iou_thres, opt.save_json, opt.augment,
             quantized=opt.quantized, a_bit=opt.a_bit, w_bit=opt.w_bit, is_gray_scale=opt.gray_scale, maxabsscaler=opt.maxabsscaler,
             shortcut_way=opt.shortcut_way)
    elif opt.task == 'benchmark':   # mAP / speed over a grid of sizes and NMS thresholds
        rows = []
        for size in (320, 416, 512, 608):
            for iou in (0.6, 0.7):
                res, _ = test(opt.cfg, opt.data, opt.weights, opt.batch_size, size, opt.conf_thres, iou, opt.save_json)
                rows.append(res)
        np.savetxt('benchmark.txt', rows, fmt='%10.4g')
