"""Host utilities of the detection engine (config parsing, layers, box algebra, NMS, weights helpers).

The reference's compression tooling (``utils/prune_utils.py``, ``utils/quantized/*``) is meant to run
*unmodified* on top of this package.  When a checkout of the reference is present (``/root/reference``
or ``$YOLO_REFERENCE_ROOT``) its ``utils`` directory is appended to this package's search path, so
``utils.prune_utils`` and ``utils.quantized.*`` resolve to the reference's files while every module
that exists here (``utils.utils``, ``utils.layers``, ``utils.torch_utils``, ``utils.parse_config``,
``utils.datasets``) resolves to the new implementation.
"""
import os as _os

_ref = _os.path.join(_os.environ.get('YOLO_REFERENCE_ROOT', '/root/reference'), 'utils')
if _os.path.isdir(_ref) and _ref not in __path__:
    __path__.append(_ref)
