"""Import the *reference* implementation (read-only, /root/reference) inside the build container.

Only used by ``tests/golden/make_golden.py`` and by container-only tests that are skipped when the
reference tree is absent (it does not exist on the GPU box).  Two I/O-only imports the reference
needs are not installed here and are stubbed: ``cv2`` (only ``setNumThreads`` is touched at import)
and ``torchvision.ops.boxes.nms`` (restated as greedy NMS: score-descending, suppress IoU > thr).

The reference's top-level module names (``models``, ``utils``) collide with this repo's own, so
``load()`` temporarily swaps the ``sys.path`` / ``sys.modules`` entries, imports the reference, puts
this repo's modules back, and returns the reference modules under private handles.
"""
import importlib
import os
import sys
import types

REF = os.environ.get('YOLO_REFERENCE_ROOT', '/root/reference')
_CLASH = ('models', 'utils', 'test', 'train', 'detect')


def available():
    return os.path.isfile(os.path.join(REF, 'models.py'))


def _greedy_nms(boxes, scores, iou_threshold):
    import torch
    order = torch.argsort(scores, descending=True, stable=True)
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    keep = []
    dead = torch.zeros(len(b), dtype=torch.bool)
    for i in range(len(b)):
        if dead[i]:
            continue
        keep.append(i)
        lt = torch.max(b[i, :2], b[:, :2])
        rb = torch.min(b[i, 2:], b[:, 2:])
        inter = (rb - lt).clamp(0).prod(1)
        dead |= inter / (area[i] + area - inter) > iou_threshold
    return order[torch.tensor(keep, dtype=torch.long)]


def _stubs():
    cv2 = types.ModuleType('cv2')
    cv2.setNumThreads = lambda n: None
    tv = types.ModuleType('torchvision')
    ops = types.ModuleType('torchvision.ops')
    boxes = types.ModuleType('torchvision.ops.boxes')
    boxes.nms = _greedy_nms
    ops.boxes = boxes
    ops.nms = _greedy_nms
    tv.ops = ops
    return {'cv2': cv2, 'torchvision': tv, 'torchvision.ops': ops, 'torchvision.ops.boxes': boxes}


_cache = {}


def load():
    """Return a namespace with the reference's ``models``, ``utils.utils`` ... modules."""
    if _cache:
        return _cache['ns']
    assert available(), 'reference tree not found at ' + REF
    saved_mods = {k: v for k, v in sys.modules.items() if k.split('.')[0] in _CLASH}
    for k in saved_mods:
        del sys.modules[k]
    saved_path = list(sys.path)
    stubs = _stubs()
    injected = [k for k in stubs if k not in sys.modules]
    for k in injected:
        sys.modules[k] = stubs[k]
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    try:
        ref_models = importlib.import_module('models')
        ref_utils = importlib.import_module('utils.utils')
        ref_torch_utils = importlib.import_module('utils.torch_utils')
        ref_parse = importlib.import_module('utils.parse_config')
        ref_layers = importlib.import_module('utils.layers')
    finally:
        sys.path[:] = saved_path
        ref_loaded = {k: v for k, v in sys.modules.items() if k.split('.')[0] in _CLASH}
        for k in ref_loaded:
            del sys.modules[k]
        sys.modules.update(saved_mods)
        for k in injected:
            sys.modules.pop(k, None)
    ns = types.SimpleNamespace(models=ref_models, utils=ref_utils, torch_utils=ref_torch_utils,
                               parse_config=ref_parse, layers=ref_layers, root=REF, _modules=ref_loaded)
    _cache['ns'] = ns
    return ns
