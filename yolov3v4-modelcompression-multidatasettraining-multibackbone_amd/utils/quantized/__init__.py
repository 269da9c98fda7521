"""Quantisation modules.

``quantized_ptq_cos`` (the post-training path the int8 HIP engine lowers) is implemented
in this package.  The QAT research quantisers of the reference (``quantized_google``, ``quantized_TPSQ`` ...) are
out of scope; when a checkout of the reference is present they still resolve to its unmodified files.
"""
import os as _os

_ref = _os.path.join(_os.environ.get('YOLO_REFERENCE_ROOT', '/root/reference'), 'utils', 'quantized')
if _os.path.isdir(_ref) and _ref not in __path__:
    __path__.append(_ref)
