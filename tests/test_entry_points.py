"""Host side of the drop-in boundary: utils.datasets, test.test, train.train, detect.detect on a small
synthetic image set (PNG files + label files generated here), CPU eager path.  The reference's own loaders need OpenCV
(absent in this image), so parity here is on documented behaviour: letterbox geometry, label transforms, collate
format, the test() return contract, checkpoint contents."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

import conftest


def test_letterbox_geometry():
    from utils.datasets import letterbox
    img = np.full((50, 100, 3), 200, np.uint8)
    out, ratio, (dw, dh) = letterbox(img, 64, auto=False)
    assert out.shape == (64, 64, 3) and ratio == (0.64, 0.64) and (dw, dh) == (0.0, 16.0)
    assert (out[:16] == 114).all() and (out[48:] == 114).all() and (out[16:48] == 200).all()
    out, ratio, (dw, dh) = letterbox(img, 128, auto=True)           # minimum rectangle: pad only to a multiple of 64
    assert out.shape == (64, 128, 3) and (dw, dh) == (0.0, 0.0)
    out, ratio, _ = letterbox(img, (64, 64), auto=False, scaleFill=True)
    assert out.shape == (64, 64, 3) and ratio == (0.64, 1.28)
    small = np.zeros((10, 20, 3), np.uint8)
    out, ratio, _ = letterbox(small, 64, auto=False, scaleup=False)  # never enlarges
    assert ratio == (1.0, 1.0) and out.shape == (64, 64, 3)


def test_dataset_items_and_collate(dataset_dir):
    from utils.datasets import LoadImagesAndLabels
    ds = LoadImagesAndLabels(str(dataset_dir / 'valid.txt'), img_size=96, batch_size=2, rect=True)
    assert len(ds) == 4 and ds.batch_shapes.shape == (2, 2) and (ds.batch_shapes % 32 == 0).all()
    ar = ds.shapes[:, 1] / ds.shapes[:, 0]
    assert (np.diff(ar) >= 0).all(), 'rect mode sorts by aspect ratio'
    batch = [ds[i] for i in range(2)]
    imgs, labels, paths, shapes = ds.collate_fn(batch)
    assert imgs.dtype == torch.uint8 and imgs.shape[0] == 2 and imgs.shape[1] == 3
    assert tuple(imgs.shape[2:]) == tuple(ds.batch_shapes[0])
    assert labels.shape[1] == 6 and set(labels[:, 0].tolist()) <= {0.0, 1.0}
    assert (labels[:, 2:] >= 0).all() and (labels[:, 2:] <= 1).all()
    # a label's box still covers its bright rectangle after the letterbox transform
    img0, lab0 = batch[0][0].numpy(), batch[0][1].numpy()
    h, w = img0.shape[1:]
    for _, cls, cx, cy, bw, bh in lab0:
        x1, x2, y1, y2 = int((cx - bw / 2) * w) + 2, int((cx + bw / 2) * w) - 2, int((cy - bh / 2) * h) + 2, int((cy + bh / 2) * h) - 2
        assert img0[:, y1:y2, x1:x2].mean() > 100
    # training mode: mosaic + flip + hsv keep the contract
    hyp = dict(hsv_h=0.0138, hsv_s=0.678, hsv_v=0.36, degrees=0, translate=0, scale=0, shear=0)
    tr = LoadImagesAndLabels(str(dataset_dir / 'train.txt'), img_size=64, batch_size=4, augment=True, hyp=hyp)
    im, lab, _, sh = tr[0]
    assert tuple(im.shape) == (3, 64, 64) and sh is None and lab.shape[1] == 6
    assert (lab[:, 2:] >= 0).all() and (lab[:, 2:] <= 1).all()


def test_train_test_detect_round_trip(dataset_dir, tiny_cfg, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    import train as train_mod
    import test as test_mod
    import detect as detect_mod
    opt = train_mod.make_parser().parse_args(['--epochs', '2', '--batch-size', '4', '--cfg', tiny_cfg, '--data', str(dataset_dir / 'synth.data'),
                                              '--img-size', '64', '64', '64', '--device', 'cpu', '--nosave', '--ema'])
    opt.local_rank = -1
    results = train_mod.train(opt, train_mod.hyp)
    assert len(results) == 7 and all(np.isfinite(results))
    assert os.path.isfile('weights/last.pt') and os.path.isfile('results.txt')
    ckpt = torch.load('weights/last.pt', map_location='cpu', weights_only=False)
    assert set(ckpt) >= {'epoch', 'best_fitness', 'training_results', 'model', 'optimizer'} and ckpt['epoch'] == 1
    assert any(k.endswith('Conv2d.weight') for k in ckpt['model'])
    # library call as the prune scripts make it: test(cfg, data, weights, ...)[0][2] is mAP
    test_mod.opt = None
    (mp, mr, m_ap, mf1, *losses), maps = test_mod.test(tiny_cfg, str(dataset_dir / 'synth.data'), 'weights/last.pt', batch_size=2, imgsz=64, plot=False)
    assert 0.0 <= m_ap <= 1.0 and maps.shape == (2,) and len(losses) == 3
    dopt = detect_mod.make_parser().parse_args(['--cfg', tiny_cfg, '--weights', 'weights/last.pt', '--source', str(dataset_dir / 'images'),
                                               '--output', 'out', '--img-size', '64', '--conf-thres', '0.001', '--device', 'cpu',
                                               '--names', str(dataset_dir / 'synth.names'), '--save-txt'])
    res = detect_mod.detect(dopt)
    assert len(res) == 12 and len(os.listdir('out')) >= 12
    # OpenCV's arithmetic exists in the device kernels only: asked for on a CPU run it raises instead of quietly using Pillow
    dopt.image_arith = 'cv2'
    with pytest.raises(NotImplementedError):
        detect_mod.detect(dopt)
    opt.image_arith, opt.epochs = 'cv2', 1
    with pytest.raises(NotImplementedError):
        train_mod.train(opt, train_mod.hyp)


def test_terminaltables_shim_and_star_imports():
    import terminaltables
    t = terminaltables.AsciiTable([['Metric', 'Before'], ['mAP', '0.5']]).table
    assert 'Metric' in t and t.count('\n') == 4
    ns = {}
    exec('from models import *\nfrom utils.utils import *\nfrom utils.datasets import *', ns)
    for name in ('Darknet', 'load_darknet_weights', 'non_max_suppression', 'compute_loss', 'LoadImages', 'LoadImagesAndLabels',
                 'scale_coords', 'xyxy2xywh', 'ap_per_class', 'plot_one_box', 'labels_to_class_weights', 'fitness', 'init_seeds',
                 'output_to_target', 'plot_images', 'strip_optimizer', 'letterbox'):
        assert name in ns, name


@pytest.mark.gpu
@pytest.mark.parametrize('device_augment', [False, True], ids=['host_loader', 'device_augment'])
def test_train_and_test_on_the_gpu_path(dataset_dir, tiny_cfg, tmp_path, monkeypatch, device_augment):
    """train.py (mixed precision) + test.test on a GPU: the HIP training and inference paths behind the entry points; by default
    (device augmentation; --host-augment switches it off) the items reach the loop as recipes and are rendered by yh_mosaic_affine_hsv."""
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    monkeypatch.chdir(tmp_path)
    import train as train_mod
    import test as test_mod
    opt = train_mod.make_parser().parse_args(['--epochs', '2', '--batch-size', '4', '--cfg', tiny_cfg, '--data', str(dataset_dir / 'synth.data'),
                                              '--img-size', '64', '64', '64', '-mpt', '--nosave'] + ([] if device_augment else ['--host-augment']))
    opt.local_rank = -1
    results = train_mod.train(opt, train_mod.hyp)
    assert len(results) == 7 and all(np.isfinite(results))
    ckpt = torch.load('weights/last.pt', map_location='cpu', weights_only=False)
    assert ckpt['epoch'] == 1
    test_mod.opt = None
    (mp, mr, m_ap, mf1, *losses), maps = test_mod.test(tiny_cfg, str(dataset_dir / 'synth.data'), 'weights/last.pt', batch_size=2, imgsz=64, plot=False)
    assert 0.0 <= m_ap <= 1.0 and all(np.isfinite(losses))


def test_ptq_entry_point_calibrates_and_saves(dataset_dir, tiny_cfg, tmp_path, monkeypatch):
    """PTQ.py (reference PTQ.py:12-131): float evaluation, calibration over the train split, quantised evaluation, PTQ.pt."""
    monkeypatch.chdir(tmp_path)
    import sys
    for name in ('utils.quantized.quantized_ptq_cos', 'utils.quantized'):   # tests/test_ptq.py may have left its stand-in there
        mod = sys.modules.get(name)
        if mod is not None and not getattr(mod, '__file__', '').startswith(conftest.PKG):
            del sys.modules[name]
    import models
    import PTQ as ptq_mod
    import test as test_mod
    torch.manual_seed(0)
    fm = models.Darknet(tiny_cfg, (64, 64))
    os.makedirs('weights', exist_ok=True)
    models.save_weights(fm, 'weights/float.weights')
    test_mod.opt = None
    opt = ptq_mod.make_parser().parse_args(['--cfg', tiny_cfg, '--data', str(dataset_dir / 'synth.data'), '--weights', 'weights/float.weights',
                                            '--batch-size', '4', '--img-size', '64', '--device', 'cpu'])
    before, after = ptq_mod.PTQ(opt)
    assert len(before[0]) == 7 and len(after[0]) == 7 and all(np.isfinite(after[0]))
    ckpt = torch.load('weights/PTQ.pt', map_location='cpu', weights_only=False)
    qm = models.Darknet(tiny_cfg, (64, 64), quantized=3, shortcut_way=1)
    qm.load_state_dict(ckpt['model'])
    scales = [v for k, v in ckpt['model'].items() if k.endswith('activation_quantizer.scale')]
    assert scales and all(float(s) > 0 for s in scales)


# ---------------------------------------------------------------------------- train.py setup pieces (reference train.py:120-262)
def _opt(**kw):
    import argparse
    base = dict(adam=False, quantized=-1)
    base.update(kw)
    return argparse.Namespace(**base)


def test_teacher_gets_its_weights(tiny_cfg, tmp_path):
    """--t_weights must reach the teacher (reference train.py:185-192): .weights and .pt both, anything else raises."""
    import models
    import train as train_module
    torch.manual_seed(3)
    src = models.Darknet(tiny_cfg)
    with torch.no_grad():
        for p in src.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    wfile, ptfile = str(tmp_path / 'teacher.weights'), str(tmp_path / 'teacher.pt')
    models.save_weights(src, wfile)
    torch.save({'model': src.state_dict()}, ptfile)
    for f in (wfile, ptfile):
        torch.manual_seed(11)   # a different initialisation than the file's
        t = train_module.load_teacher(tiny_cfg, f, torch.device('cpu'))
        assert not t.training
        for (k, a), (_, b) in zip(t.named_parameters(), src.named_parameters()):
            assert torch.equal(a, b), k
    with pytest.raises(Exception, match='teacher weights'):
        train_module.load_teacher(tiny_cfg, '', torch.device('cpu'))


def test_optimizer_recipe_matches_reference(tiny_cfg):
    import models
    import train as train_module
    model = models.Darknet(tiny_cfg)
    hyp = dict(train_module.hyp)
    sgd = train_module.build_optimizer(model, _opt(), hyp, torch.device('cpu'))
    assert isinstance(sgd, torch.optim.SGD) and sgd.param_groups[0]['lr'] == hyp['lr0'] and sgd.param_groups[0]['nesterov']
    assert [g['weight_decay'] for g in sgd.param_groups] == [0, hyp['weight_decay'], 0]
    names = dict(model.named_parameters())
    n_bias = sum('.bias' in k for k in names)
    n_convw = sum('Conv2d.weight' in k and '.bias' not in k for k in names)
    assert [len(g['params']) for g in sgd.param_groups] == [len(names) - n_bias - n_convw, n_convw, n_bias]
    for o in (_opt(adam=True), _opt(quantized=3)):   # train.py:135: Adam at lr0 * 0.005 for --adam or any quantised graph
        adam = train_module.build_optimizer(model, o, hyp, torch.device('cpu'))
        assert isinstance(adam, torch.optim.Adam) and abs(adam.param_groups[0]['lr'] - hyp['lr0'] * 0.005) < 1e-12


@pytest.mark.skipif(not os.path.isdir('/root/reference/utils'), reason='needs the reference checkout (utils.prune_utils)')
@pytest.mark.parametrize('mode', [0, 1, 2])
def test_sparsity_layer_selection_per_prune_mode(mode):
    """--prune 0 / 1 / 2 pick their BN layers with parse_module_defs / parse_module_defs2 (5 returns) / parse_module_defs4."""
    import models
    import train as train_module
    import utils.prune_utils as pu
    defs = models.Darknet(os.path.join(conftest.PKG, 'cfg', 'yolov3', 'yolov3.cfg')).module_defs
    idx = train_module.sparsity_layers(mode, defs)
    want = {0: lambda: pu.parse_module_defs(defs)[2], 1: lambda: pu.parse_module_defs2(defs)[2], 2: lambda: pu.parse_module_defs4(defs)[2]}[mode]()
    assert list(idx) == list(want) and len(idx) > 0


def test_model_with_engine_attrs_deepcopies_and_pickles(tiny_cfg, tmp_path):
    """deepcopy / torch.save(model) after a HIP forward: the engine attributes (ctypes handles) must not travel."""
    import copy
    import ctypes
    import models
    model = models.Darknet(tiny_cfg)
    handle = ctypes.CDLL(None)   # stands in for the engines: ctypes objects refuse to pickle
    model.__dict__['_hip_engine'] = handle
    model.__dict__['_hip_train_engine'] = handle
    clone = copy.deepcopy(model)
    assert clone.__dict__['_hip_engine'] is None and clone.__dict__['_hip_train_engine'] is None
    path = str(tmp_path / 'whole_model.pt')
    torch.save(model, path)
    back = torch.load(path, weights_only=False)
    assert back.__dict__['_hip_engine'] is None and back.__dict__['_hip_train_engine'] is None
    assert model.__dict__['_hip_engine'] is handle   # the live model keeps its engines
    for (k, a), (_, b) in zip(back.named_parameters(), model.named_parameters()):
        assert torch.equal(a, b), k
