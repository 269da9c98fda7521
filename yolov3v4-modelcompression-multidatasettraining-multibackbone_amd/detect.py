"""Detection entry point (reference detect.py:9-178): images in ``--source`` -> boxes drawn / written to ``--output``.

Same options and flow as the reference (Darknet from a cfg, .pt or darknet weights, letterboxed frames ``/ 256``,
``non_max_suppression`` at conf 0.3 / iou 0.6 with ``multi_label=False``, boxes rescaled to the original frame), on the
HIP path when the device is a GPU.  Image files only: video / camera sources need OpenCV, which this image lacks.
"""
import argparse
import os
import random
import shutil
import time
from pathlib import Path

import numpy as np
import torch

from models import Darknet, attempt_download, load_darknet_weights
from utils import torch_utils
from utils.datasets import LoadImages
from utils.utils import load_classes, non_max_suppression, plot_one_box, scale_coords, xyxy2xywh


def detect(opt, save_img=True):
    imgsz = opt.img_size
    out, source, weights = opt.output, opt.source, opt.weights
    if source == '0' or source.startswith(('rtsp', 'http')) or source.endswith('.txt'):
        raise NotImplementedError('camera / stream sources need OpenCV, which this image does not ship')
    device = torch_utils.select_device(opt.device)
    if os.path.exists(out):
        shutil.rmtree(out)
    os.makedirs(out)

    model = Darknet(opt.cfg, imgsz, quantized=opt.quantized, quantizer_output=opt.quantizer_output, layer_idx=opt.layer_idx,
                    reorder=opt.reorder, TN=opt.TN, TM=opt.TM, a_bit=opt.a_bit, w_bit=opt.w_bit, FPGA=opt.FPGA,
                    is_gray_scale=opt.gray_scale, maxabsscaler=opt.maxabsscaler, shortcut_way=opt.shortcut_way)
    if weights:
        attempt_download(weights)
        if weights.endswith('.pt'):
            model.load_state_dict(torch.load(weights, map_location='cpu', weights_only=False)['model'], strict=False)
        else:
            load_darknet_weights(model, weights)
    model.to(device).eval()
    model.hip_return_raw = False      # only model(img)[0] is read below (reference detect.py:104): no copies of the raw head maps

    device_letterbox = device.type == 'cuda' and not getattr(opt, 'host_letterbox', False)
    if getattr(opt, 'image_arith', None) == 'cv2' and not device_letterbox:
        raise NotImplementedError("--image-arith cv2 (OpenCV's arithmetic) exists in the device letterbox only: it needs a GPU and no "
                                  "--host-letterbox; the host loader resizes with Pillow")
    dataset = LoadImages(source, img_size=imgsz, is_gray_scale=opt.gray_scale, rect=opt.rect, host_letterbox=not device_letterbox)
    names = load_classes(opt.names) if opt.names and os.path.isfile(opt.names) else [str(i) for i in range(1000)]
    rng = random.Random(0)
    colors = [[rng.randint(0, 255) for _ in range(3)] for _ in range(len(names))]

    t0 = time.time()
    results = []
    for path, img, im0, _ in dataset:
        if device_letterbox:
            # the decoded frame goes to the GPU as it is; resize + border + x / 256 + HWC -> CHW run there (csrc/preprocess.hip),
            # bit-identical to the loader's host letterbox (tests/test_preprocess.py)
            from engine.preprocess import letterbox_to_device
            x, _, _ = letterbox_to_device(im0, imgsz, device, auto=opt.rect, maxabsscaler=opt.maxabsscaler,
                                          arith=getattr(opt, 'image_arith', None))
        else:
            x = torch.from_numpy(img).to(device).float() / 256.0      # uint8 -> [0, 1) like detect.py:101
            if opt.maxabsscaler:
                x = x * 2 - 1
        if x.ndimension() == 3:
            x = x.unsqueeze(0)
        t1 = torch_utils.time_synchronized()
        with torch.no_grad():
            if opt.augment or not hasattr(model, 'hip_detect'):
                pred = model(x, augment=opt.augment)[0]
                t2 = torch_utils.time_synchronized()
                pred = non_max_suppression(pred.float(), opt.conf_thres, opt.iou_thres, multi_label=False, classes=opt.classes,
                                           agnostic=opt.agnostic_nms)
            else:      # forward + NMS as one engine call (models.Darknet.hip_detect: same boxes, no decoded tensor in between)
                pred = model.hip_detect(x, opt.conf_thres, opt.iou_thres, multi_label=False, classes=opt.classes,
                                        agnostic=opt.agnostic_nms)
                t2 = torch_utils.time_synchronized()
        for det in pred:
            save_path = str(Path(out) / Path(path).name)
            s = '%gx%g ' % tuple(x.shape[2:])
            im0 = np.array(im0 if im0.shape[2] == 3 else np.repeat(im0, 3, 2))   # own, writable copy for drawing
            gn = torch.tensor(im0.shape)[[1, 0, 1, 0]].float()
            n_det = 0
            if det is not None and len(det):
                det = det.clone()
                det[:, :4] = scale_coords(x.shape[2:], det[:, :4], im0.shape).round()
                for c in det[:, -1].unique():
                    s += '%g %ss, ' % ((det[:, -1] == c).sum(), names[int(c)])
                for *xyxy, conf, cls in det.tolist():
                    n_det += 1
                    if opt.save_txt:
                        xywh = (xyxy2xywh(torch.tensor(xyxy).view(1, 4)) / gn).view(-1).tolist()
                        with open(save_path[:save_path.rfind('.')] + '.txt', 'a') as f:
                            f.write(('%g ' * 5 + '\n') % (cls, *xywh))
                    if save_img:
                        plot_one_box(xyxy, im0, label='%s %.2f' % (names[int(cls)], conf), color=colors[int(cls) % len(colors)])
            print('%sDone. (%.3fs)' % (s, t2 - t1))
            results.append((path, n_det))
            if save_img:
                from PIL import Image
                Image.fromarray(im0).save(save_path)
    print('Results saved to %s' % os.path.join(os.getcwd(), out))
    print('Done. (%.3fs)' % (time.time() - t0))
    return results


def make_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--cfg', type=str, default='cfg/yolov3/yolov3.cfg', help='*.cfg path')
    parser.add_argument('--names', type=str, default='data/coco.names', help='*.names path')
    parser.add_argument('--weights', type=str, default='', help='weights path (.pt / darknet); empty = random init')
    parser.add_argument('--source', type=str, default='data/samples', help='image file or folder')
    parser.add_argument('--output', type=str, default='output', help='output folder')
    parser.add_argument('--img-size', type=int, default=512, help='inference size (pixels)')
    parser.add_argument('--conf-thres', type=float, default=0.3, help='object confidence threshold')
    parser.add_argument('--iou-thres', type=float, default=0.6, help='IOU threshold for NMS')
    parser.add_argument('--fourcc', type=str, default='mp4v', help='output video codec (unused: no video I/O here)')
    parser.add_argument('--device', default='', help='device id (i.e. 0 or 0,1) or cpu')
    parser.add_argument('--rect', action='store_true', help='rectangular detecting')
    parser.add_argument('--host-letterbox', action='store_true', help='letterbox on the host (default on a GPU: on the device)')
    parser.add_argument('--image-arith', choices=['pillow', 'cv2'], default=None,
                        help="resize arithmetic of the device letterbox: 'pillow' = the host loader's, 'cv2' = the reference's cv2.resize "
                             "restated; default: $YOLO_IMAGE_ARITH or pillow")
    parser.add_argument('--view-img', action='store_true', help='display results (unused: headless)')
    parser.add_argument('--save-txt', action='store_true', help='save results to *.txt')
    parser.add_argument('--classes', nargs='+', type=int, help='filter by class')
    parser.add_argument('--agnostic-nms', action='store_true', help='class-agnostic NMS')
    parser.add_argument('--augment', action='store_true', help='augmented inference')
    parser.add_argument('--quantized', type=int, default=-1, help='quantization way')
    parser.add_argument('--shortcut_way', type=int, default=1, help='--shortcut quantization way')
    parser.add_argument('--a_bit', type=int, default=8, help='a-bit')
    parser.add_argument('--w_bit', type=int, default=8, help='w-bit')
    parser.add_argument('--FPGA', action='store_true', help='FPGA')
    parser.add_argument('--quantizer_output', action='store_true', help='quantizer output')
    parser.add_argument('--layer_idx', type=int, default=-1, help='output')
    parser.add_argument('--reorder', action='store_true', help='reorder')
    parser.add_argument('--TN', type=int, default=32, help='TN')
    parser.add_argument('--TM', type=int, default=32, help='TM')
    parser.add_argument('--gray-scale', action='store_true', help='gray scale training')
    parser.add_argument('--maxabsscaler', '-mas', action='store_true', help='standardise input to (-1, 1)')
    return parser


if __name__ == '__main__':
    opt = make_parser().parse_args()
    print(opt)
    with torch.no_grad():
        detect(opt)
